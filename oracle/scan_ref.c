/* CPU ORACLE (test infrastructure only) - plain C restatement of the selective-scan recurrence
 * and of the MoE dispatch plan.  Built by oracle/build_c.py into oracle/_build/liboracle_ref.so and
 * checked against the golden vectors captured from the reference (tests/test_oracle_c_cpu.py).
 *
 * scan:  SelectiveLinearAttention._ssm_pytorch_scan_recurrent, /root/reference/src/model/core.py:337-353
 *        (token-major layout: delta [B,L,h], Bt/C/y [B,L,h*N]); backward = the adjoint in
 *        SURVEY.md section 8a row S-bwd (the reference relies on autograd).
 * plan:  the K x E dispatch loop of AdaptiveExpertSystem.forward, core.py:547-591, emitted in the
 *        canonical expert-major order (capacity is consumed per expert, so the orders are equivalent).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* core.py:339-349 : A = -exp(A_log); a = exp(delta*A); s = a*s + Bt; y = C*s */
void oracle_scan_fwd_f32(const float *delta, const float *A_log, const float *Bt, const float *C, const float *h0,
                         float *y, float *h_last, int64_t B, int64_t L, int64_t h, int64_t N) {
  const int64_t Dn = h * N;
  for (int64_t b = 0; b < B; ++b)
    for (int64_t c = 0; c < Dn; ++c) {
      const float A = -expf(A_log[c]);
      float s = h0 ? h0[b * Dn + c] : 0.f;
      for (int64_t t = 0; t < L; ++t) {
        const float a = expf(delta[(b * L + t) * h + c / N] * A);
        s = a * s + Bt[(b * L + t) * Dn + c];
        y[(b * L + t) * Dn + c] = C[(b * L + t) * Dn + c] * s;
      }
      if (h_last) h_last[b * Dn + c] = s;
    }
}

/* adjoint: dC = g*s_t ; lambda_t = g*C_t + a_{t+1}*lambda_{t+1} ; dBt = lambda ; da_t = lambda_t*s_{t-1}
 * d_delta[b,t,head] = sum_n da*a*A ; dA_log[c] = sum_{b,t} da*a*delta*A   (states kept in a scratch column) */
void oracle_scan_bwd_f32(const float *delta, const float *A_log, const float *Bt, const float *C, const float *dy,
                         const float *h0, float *d_delta, float *dA_log, float *dBt, float *dC, int64_t B, int64_t L,
                         int64_t h, int64_t N) {
  const int64_t Dn = h * N;
  double *st = (double *)malloc(sizeof(double) * (size_t)(L + 1));
  memset(d_delta, 0, sizeof(float) * (size_t)(B * L * h));
  for (int64_t c = 0; c < Dn; ++c) {
    const double A = -exp((double)A_log[c]);
    double dA = 0.0;
    for (int64_t b = 0; b < B; ++b) {
      st[0] = h0 ? h0[b * Dn + c] : 0.0;
      for (int64_t t = 0; t < L; ++t)
        st[t + 1] = exp((double)delta[(b * L + t) * h + c / N] * A) * st[t] + Bt[(b * L + t) * Dn + c];
      double lam_next = 0.0, a_next = 0.0;
      for (int64_t t = L - 1; t >= 0; --t) {
        const int64_t i = (b * L + t) * Dn + c;
        const double dl = delta[(b * L + t) * h + c / N], a = exp(dl * A), g = dy[i];
        const double lam = g * C[i] + (t + 1 < L ? a_next * lam_next : 0.0);
        dC[i] = (float)(g * st[t + 1]);
        dBt[i] = (float)lam;
        const double q = lam * st[t] * a * A;
        d_delta[(b * L + t) * h + c / N] += (float)q;
        dA += q * dl;
        lam_next = lam;
        a_next = a;
      }
    }
    dA_log[c] = (float)dA;
  }
  free(st);
}

typedef struct { float w; int32_t tok; } cand_t;
static int cand_cmp(const void *pa, const void *pb) {
  const cand_t *a = (const cand_t *)pa, *b = (const cand_t *)pb;
  if (a->w != b->w) return a->w > b->w ? -1 : 1; /* larger gate weight first (core.py:580-581) */
  return a->tok < b->tok ? -1 : (a->tok > b->tok);  /* ties: lowest token */
}
static int i32_cmp(const void *a, const void *b) { return *(const int32_t *)a - *(const int32_t *)b; }

/* idx [S,K] int32, w [S,K] f32, active [E] u8 or NULL, capacity <= 0 = unlimited.
 * Outputs: offsets [E+1], row_token/row_k [S*K] (first offsets[E] used), slot_of [S,K] (-1 dropped). */
void oracle_moe_plan(const int32_t *idx, const float *w, const uint8_t *active, int64_t capacity, int32_t *offsets,
                     int32_t *row_token, int32_t *row_k, int32_t *slot_of, int64_t S, int64_t E, int64_t K) {
  cand_t *cand = (cand_t *)malloc(sizeof(cand_t) * (size_t)(S > 0 ? S : 1));
  int32_t *kept = (int32_t *)malloc(sizeof(int32_t) * (size_t)(S > 0 ? S : 1));
  for (int64_t i = 0; i < S * K; ++i) slot_of[i] = -1;
  int32_t rows = 0;
  for (int64_t e = 0; e < E; ++e) {
    offsets[e] = rows;
    int64_t load = 0;
    if (active && !active[e]) continue;                         /* core.py:552 */
    for (int64_t k = 0; k < K; ++k) {                           /* capacity is consumed k-major, :547 */
      int64_t n = 0;
      for (int64_t s = 0; s < S; ++s)
        if (idx[s * K + k] == e) { cand[n].w = w[s * K + k]; cand[n].tok = (int32_t)s; ++n; }   /* :556-561 */
      int64_t keep = n;
      if (capacity > 0) {
        const int64_t rem = capacity - load;                    /* :568 */
        if (rem <= 0) continue;                                 /* :570 */
        if (keep > rem) keep = rem;                             /* :576 */
      }
      if (keep < n) qsort(cand, (size_t)n, sizeof(cand_t), cand_cmp);   /* :578-582 keep the top `keep` */
      for (int64_t i = 0; i < keep; ++i) kept[i] = cand[i].tok;
      qsort(kept, (size_t)keep, sizeof(int32_t), i32_cmp);      /* canonical: ascending token */
      for (int64_t i = 0; i < keep; ++i) {
        slot_of[(int64_t)kept[i] * K + k] = rows;
        row_token[rows] = kept[i];
        row_k[rows] = (int32_t)k;
        ++rows;
      }
      load += keep;                                             /* :590 */
    }
  }
  offsets[E] = rows;
  free(cand);
  free(kept);
}
