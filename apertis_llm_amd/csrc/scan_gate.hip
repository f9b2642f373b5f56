// Selective scan with the post-scan skip + gate fused in (gfx950), forward and backward, and the single-token decode step.
//
// Reference: SelectiveLinearAttention.forward, /root/reference/src/model/core.py:388-396 -
//   s_t = exp(delta_t*A)*s_{t-1} + Bt_t,  y_t = C_t*s_t                      (:337-353, the recurrence)
//   out_t = (y_t + D*xc_t) * silu(z_t)                                        (:395-396)
// The stand-alone pair (selective_scan.hip + apertis_ssm_gate_*) writes y in fp32, reads it back for the gate and does
// the same with dy in the backward: half of the scan's HBM bytes.  Here y never reaches HBM: the forward writes the gated
// output in the activation dtype, the backward recomputes y from the states it rebuilds anyway.
//
// Algorithmic bytes per token (T = B*L, e = bytes of the activation dtype; SURVEY.md 8(d) "fused epilogue variant"):
//   forward   Dn*e*(Bt + C + xc + z + out) + 4h          = 5*Dn*e + 4h
//   backward  Dn*e*(Bt + C + xc + z + dout) reads + Dn*e*(dBt + dC + dxc + dz) writes + 8h = 9*Dn*e + 8h
//
// Structure: work-group = (batch, chunk of LT tokens, 64-channel tile); lane = channel, wave = token segment; tiles are
// staged HBM -> LDS in whole row segments (16 B per lane when the slices allow), each thread walks its LDS column.
// Two ways to get a chunk's carry-in:
//   mode 0  two launches: a state pass writes every chunk's aggregate, the replay pass composes its carry from them
//           (the scheme of selective_scan.hip; the first pass reads Bt - resp. C, dout, z - a second time);
//   mode 1  ONE launch: work-groups take their item from a ticket counter in chunk-major order, compute the chunk
//           aggregate from the Bt tile first, PUBLISH it (8-byte {epoch, value} granules, one agent-scope store each:
//           cdna_hip_programming.md Guideline 16, form R2) and gather the aggregates of the earlier chunks while their
//           own C / xc / z loads are still in flight.  A work-group only ever waits for lower tickets, which belong to
//           work-groups that have already started and never wait for a higher one: progress does not depend on
//           residency.  The composition order is fixed (all predecessors, left to right), so the result is
//           bit-identical to mode 0 and run-to-run.  Waits are bounded; a timeout sets the workspace's error word.
#include "scan_common.h"

namespace {

typedef unsigned long long gran_t;
__device__ __forceinline__ void gran_store(gran_t *p, uint32_t epoch, float v) {
  __hip_atomic_store(p, ((gran_t)epoch << 32) | (gran_t)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ gran_t gran_load(const gran_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// head of the look-back workspace (64 bytes), then the granules [B*ctiles][nchunks][64 lanes][2]
struct GateWsHead { unsigned ctr[2]; int err; int pad[13]; };

__device__ __forceinline__ float sigmoid_g(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float silu_g(float x) { return x * sigmoid_g(x); }
__device__ __forceinline__ float silu_grad_g(float x) { float s = sigmoid_g(x); return s * (1.f + x * (1.f - s)); }

// global -> registers half of stage_in (the loads stay in flight until stage_regs_store)
template <int VB, int ROWB, int LT, int NTH>
struct TileRegs {
  typedef typename vec_bytes<VB>::type V;
  static constexpr int CPR = ROWB / VB, TOTAL = LT * CPR, ITERS = (TOTAL + NTH - 1) / NTH;
  V r[ITERS];
  __device__ __forceinline__ void load(const char *g, int64_t rsb, int rows_valid, int bytes_valid, int tid) {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      int idx = tid + it * NTH;
      int row = idx / CPR, cb = (idx % CPR) * VB;
      bool ok = idx < TOTAL && row < rows_valid && cb < bytes_valid;
      r[it] = ok ? *reinterpret_cast<const V *>(g + (int64_t)row * rsb + cb) : zero_vec<VB>();
    }
  }
  __device__ __forceinline__ void store(char *lds, int tid) const {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      int idx = tid + it * NTH;
      if (idx < TOTAL) *reinterpret_cast<V *>(lds + idx * VB) = r[it];
    }
  }
};

// Gather (mode 1): compose the published aggregates of chunks [s0, s1) in composition order into (P, S).
// forward: j ascending; reverse: j descending.  Spins (bounded) until every granule carries this launch's epoch.
__device__ __forceinline__ void gather_published(const gran_t *gbase, int s0, int s1, bool reverse, int lane, bool chan_ok,
                                                 uint32_t epoch, int *err, float &P, float &S) {
  constexpr int QMAX = 4;
  P = 1.f; S = 0.f;
  for (int j0 = 0; j0 < s1 - s0; j0 += QMAX) {
    gran_t gp[QMAX], gs[QMAX];
    const int nb = min(QMAX, s1 - s0 - j0);
    unsigned spins = 0;
    while (true) {
      bool ok = true;
#pragma unroll
      for (int u = 0; u < QMAX; ++u)
        if (u < nb && chan_ok) {
          const int j = reverse ? (s1 - 1 - j0 - u) : (s0 + j0 + u);
          gp[u] = gran_load(gbase + ((int64_t)j * TC + lane) * 2 + 0);
          gs[u] = gran_load(gbase + ((int64_t)j * TC + lane) * 2 + 1);
        }
#pragma unroll
      for (int u = 0; u < QMAX; ++u)
        if (u < nb && chan_ok) ok = ok && (uint32_t)(gp[u] >> 32) == epoch && (uint32_t)(gs[u] >> 32) == epoch;
      if (__all(ok)) break;
      if (++spins > (1u << 18)) { if (lane == 0) atomicOr(err, 1); break; }
      __builtin_amdgcn_s_sleep(4);
    }
#pragma unroll
    for (int u = 0; u < QMAX; ++u)
      if (u < nb && chan_ok) {
        const float pj = __uint_as_float((uint32_t)gp[u]), sj = __uint_as_float((uint32_t)gs[u]);
        S = fmaf(pj, S, sj);
        P *= pj;
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// forward: out = (C*s + D*xc) * silu(z)
template <typename T, int VB, int LT, int NS, int MODE>
__global__ void __launch_bounds__(TC * NS)
scan_gate_fwd_k(const float *__restrict__ dlt, const float *__restrict__ A_log, const T *__restrict__ Bt, int64_t bt_rs,
                const T *__restrict__ C, int64_t c_rs, const T *__restrict__ xc, int64_t xc_rs, const T *__restrict__ z,
                int64_t z_rs, const float *__restrict__ Dv, const float *__restrict__ h0, const float2 *__restrict__ agg,
                GateWsHead *__restrict__ head, gran_t *__restrict__ gran, uint32_t epoch, float *__restrict__ h_in,
                float *__restrict__ h_last, T *__restrict__ out, int64_t out_rs, ScanDims d, int64_t nch64, int ctiles) {
  constexpr int ROWB = TC * sizeof(T);
  constexpr int TS = LT / NS;
  constexpr int NTH = TC * NS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T *bt = reinterpret_cast<T *>(smem);
  T *cc = bt + LT * TC;
  T *xx = cc + LT * TC;
  T *zz = xx + LT * TC;
  float *dl = reinterpret_cast<float *>(zz + LT * TC);
  float2 *segs = reinterpret_cast<float2 *>(dl + LT * d.HT);
  float2 *lk = segs + NS * TC;
  int *item_s = reinterpret_cast<int *>(lk + NS * TC);

  const int tid = threadIdx.x, lane = tid & 63, seg = tid >> 6;
  int chunk, ct, b;
  if constexpr (MODE == 1) {
    if (tid == 0) {
      const unsigned t = atomicAdd(&head->ctr[epoch & 1], 1u);
      if (t == 0) atomicExch(&head->ctr[(epoch + 1) & 1], 0u);   // the next launch's counter (its last user has finished)
      *item_s = (int)t;
    }
    __syncthreads();
    const int item = *item_s, bct_n = (int)d.B * ctiles;
    chunk = item / bct_n;
    const int bct = item - chunk * bct_n;
    b = bct / ctiles;
    ct = bct - b * ctiles;
  } else {
    chunk = blockIdx.x; ct = blockIdx.y; b = blockIdx.z;
  }
  const int c0 = ct * TC, c = c0 + lane;
  const int64_t t0 = (int64_t)chunk * LT;
  const int rows_valid = (int)min((int64_t)LT, d.L - t0);
  const int ch_valid = (int)min((int64_t)TC, d.Dn - c0);
  const int64_t tok0 = (int64_t)b * d.L + t0;
  const bool chan_ok = c < d.Dn;

  stage_in<VB, ROWB, LT, NTH>(reinterpret_cast<char *>(bt), reinterpret_cast<const char *>(Bt + tok0 * bt_rs + c0),
                              bt_rs * sizeof(T), rows_valid, ch_valid * (int)sizeof(T), tid);
  stage_delta<LT, NTH>(dl, dlt, tok0, rows_valid, c0 >> d.log2N, (int)d.h, d.HT, d.softplus, tid);
  const float A2 = chan_ok ? -expf(A_log[c]) * LOG2E_F : 0.f;
  const float Dc = chan_ok ? Dv[c] : 0.f;
  // the other three tiles: loads issued now, parked in registers while the aggregates are formed (and, in mode 1,
  // published and gathered)
  TileRegs<VB, ROWB, LT, NTH> rc, rx, rz;
  rc.load(reinterpret_cast<const char *>(C + tok0 * c_rs + c0), c_rs * sizeof(T), rows_valid, ch_valid * (int)sizeof(T), tid);
  rx.load(reinterpret_cast<const char *>(xc + tok0 * xc_rs + c0), xc_rs * sizeof(T), rows_valid, ch_valid * (int)sizeof(T), tid);
  rz.load(reinterpret_cast<const char *>(z + tok0 * z_rs + c0), z_rs * sizeof(T), rows_valid, ch_valid * (int)sizeof(T), tid);
  __syncthreads();   // Bt and delta tiles are in LDS

  float a[TS];
  float P = 1.f, S = 0.f;
  const int hh = lane >> d.log2N;
#pragma unroll
  for (int i = 0; i < TS; ++i) {
    const int t = seg * TS + i;
    a[i] = __builtin_amdgcn_exp2f(dl[t * d.HT + hh] * A2);
    S = fmaf(a[i], S, to_f32(bt[t * TC + lane]));
    P *= a[i];
  }
  segs[seg * TC + lane] = make_float2(P, S);
  __syncthreads();

  // carry entering the chunk: composed left to right from the aggregates of chunks [0, chunk); the NS waves split
  // the range, partials meet in lk
  {
    const int q = (chunk + NS - 1) / NS;
    const int s0 = min(seg * q, chunk), s1 = min(s0 + q, chunk);
    float Pw = 1.f, Sw = 0.f;
    if constexpr (MODE == 1) {
      gran_t *gbase = gran + ((int64_t)(b * ctiles + ct) * d.nchunks) * TC * 2;
      if (seg == 0 && chunk + 1 < d.nchunks) {   // publish this chunk's aggregate first (the last chunk has no reader)
        float Pc = P, Sc = S;
#pragma unroll
        for (int s = 1; s < NS; ++s) { const float2 r = segs[s * TC + lane]; Sc = fmaf(r.x, Sc, r.y); Pc *= r.x; }
        if (chan_ok) {
          gran_store(gbase + ((int64_t)chunk * TC + lane) * 2 + 0, epoch, Pc);
          gran_store(gbase + ((int64_t)chunk * TC + lane) * 2 + 1, epoch, Sc);
        }
      }
      gather_published(gbase, s0, s1, false, lane, chan_ok, epoch, &head->err, Pw, Sw);
    } else {
      if (chan_ok) {
        const int64_t base = (int64_t)b * d.nchunks * d.Dn + c;
        for (int j = s0; j < s1; ++j) { const float2 r = agg[base + (int64_t)j * d.Dn]; Sw = fmaf(r.x, Sw, r.y); Pw *= r.x; }
      }
    }
    lk[seg * TC + lane] = make_float2(Pw, Sw);
  }
  rc.store(reinterpret_cast<char *>(cc), tid);
  rx.store(reinterpret_cast<char *>(xx), tid);
  rz.store(reinterpret_cast<char *>(zz), tid);
  __syncthreads();

  float hcar = (h0 && chan_ok) ? h0[(int64_t)b * d.Dn + c] : 0.f;
#pragma unroll
  for (int s = 0; s < NS; ++s) { const float2 r = lk[s * TC + lane]; hcar = fmaf(r.x, hcar, r.y); }
  for (int s = 0; s < seg; ++s) { const float2 r = segs[s * TC + lane]; hcar = fmaf(r.x, hcar, r.y); }
  // state entering every 64-token block, saved for the backward
  if ((seg * TS) % 64 == 0 && chan_ok) {
    const int64_t j64 = (int64_t)chunk * (LT / 64) + (seg * TS) / 64;
    if (j64 < nch64) h_in[((int64_t)b * nch64 + j64) * d.Dn + c] = hcar;
  }
  float hst = hcar;
#pragma unroll
  for (int i = 0; i < TS; ++i) {
    const int t = seg * TS + i;
    hst = fmaf(a[i], hst, to_f32(bt[t * TC + lane]));
    const float yv = to_f32(cc[t * TC + lane]) * hst;
    const float dx = Dc * to_f32(xx[t * TC + lane]);
    const float v = yv + dx;
    cc[t * TC + lane] = from_f32<T>(v * silu_g(to_f32(zz[t * TC + lane])));   // in place: own column only
  }
  if (h_last && chunk == d.nchunks - 1 && seg == NS - 1 && chan_ok) h_last[(int64_t)b * d.Dn + c] = hst;
  __syncthreads();
  stage_out<VB, ROWB, LT, NTH>(reinterpret_cast<const char *>(cc), reinterpret_cast<char *>(out + tok0 * out_rs + c0),
                               out_rs * sizeof(T), rows_valid, ch_valid * (int)sizeof(T), tid);
}

// ---------------------------------------------------------------------------------------------------------------
// backward, mode 0 first launch: reverse chunk aggregates (P = prod a, M = mu at chunk start from zero) with
//   u_t = dout_t*silu(z_t)*C_t,  mu_t = a_t*(u_t + mu_{t+1})
template <typename T, int VB, int LT, int NS>
__global__ void __launch_bounds__(TC * NS)
scan_gate_bwd_state_k(const float *__restrict__ dlt, const float *__restrict__ A_log, const T *__restrict__ C, int64_t c_rs,
                      const T *__restrict__ z, int64_t z_rs, const T *__restrict__ dout, int64_t do_rs,
                      float2 *__restrict__ agg, ScanDims d) {
  constexpr int ROWB = TC * sizeof(T);
  constexpr int TS = LT / NS;
  constexpr int NTH = TC * NS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T *cc = reinterpret_cast<T *>(smem);
  T *zz = cc + LT * TC;
  T *gg = zz + LT * TC;
  float *dl = reinterpret_cast<float *>(gg + LT * TC);
  float2 *segs = reinterpret_cast<float2 *>(dl + LT * d.HT);
  const int tid = threadIdx.x, lane = tid & 63, seg = tid >> 6;
  const int chunk = blockIdx.x, ct = blockIdx.y, b = blockIdx.z;
  const int c0 = ct * TC, c = c0 + lane;
  const int64_t t0 = (int64_t)chunk * LT;
  const int rows_valid = (int)min((int64_t)LT, d.L - t0);
  const int ch_valid = (int)min((int64_t)TC, d.Dn - c0);
  const int64_t tok0 = (int64_t)b * d.L + t0;
  stage_in<VB, ROWB, LT, NTH>(reinterpret_cast<char *>(cc), reinterpret_cast<const char *>(C + tok0 * c_rs + c0),
                              c_rs * sizeof(T), rows_valid, ch_valid * (int)sizeof(T), tid);
  stage_in<VB, ROWB, LT, NTH>(reinterpret_cast<char *>(zz), reinterpret_cast<const char *>(z + tok0 * z_rs + c0),
                              z_rs * sizeof(T), rows_valid, ch_valid * (int)sizeof(T), tid);
  stage_in<VB, ROWB, LT, NTH>(reinterpret_cast<char *>(gg), reinterpret_cast<const char *>(dout + tok0 * do_rs + c0),
                              do_rs * sizeof(T), rows_valid, ch_valid * (int)sizeof(T), tid);
  stage_delta<LT, NTH>(dl, dlt, tok0, rows_valid, c0 >> d.log2N, (int)d.h, d.HT, d.softplus, tid);
  const float A2 = c < d.Dn ? -expf(A_log[c]) * LOG2E_F : 0.f;
  __syncthreads();
  float P = 1.f, M = 0.f;
  const int hh = lane >> d.log2N;
#pragma unroll
  for (int i = TS - 1; i >= 0; --i) {
    const int t = seg * TS + i;
    const float av = __builtin_amdgcn_exp2f(dl[t * d.HT + hh] * A2);
    const float dv = to_f32(gg[t * TC + lane]) * silu_g(to_f32(zz[t * TC + lane]));
    const float u = dv * to_f32(cc[t * TC + lane]);
    M = av * (u + M);
    P *= av;
  }
  segs[seg * TC + lane] = make_float2(P, M);
  __syncthreads();
  if (seg == 0 && c < d.Dn) {
    float2 q = segs[(NS - 1) * TC + lane];
    float Pt = q.x, Mt = q.y;
#pragma unroll
    for (int s = NS - 2; s >= 0; --s) { const float2 r = segs[s * TC + lane]; Mt = fmaf(r.x, Mt, r.y); Pt *= r.x; }
    agg[((int64_t)b * d.nchunks + chunk) * d.Dn + c] = make_float2(Pt, Mt);
  }
}

// backward replay (mode 0: carry from agg; mode 1: single launch, chunks taken right to left from the ticket counter)
template <typename T, int VB, int LT, int NS, int MODE>
__global__ void __launch_bounds__(TC * NS)
scan_gate_bwd_k(const float *__restrict__ dlt, const float *__restrict__ A_log, const T *__restrict__ Bt, int64_t bt_rs,
                const T *__restrict__ C, int64_t c_rs, const T *__restrict__ xc, int64_t xc_rs, const T *__restrict__ z,
                int64_t z_rs, const float *__restrict__ Dv, const T *__restrict__ dout, int64_t do_rs,
                const float *__restrict__ h_in, const float2 *__restrict__ agg, GateWsHead *__restrict__ head,
                gran_t *__restrict__ gran, uint32_t epoch, T *__restrict__ dBt, int64_t dbt_rs, T *__restrict__ dC,
                int64_t dc_rs, int64_t store_w, T *__restrict__ dxc, int64_t dxc_rs, T *__restrict__ dz, int64_t dz_rs,
                float *__restrict__ d_dlt, float *__restrict__ part, ScanDims d, int ctiles) {
  constexpr int ROWB = TC * sizeof(T);
  constexpr int TS = LT / NS;
  constexpr int NTH = TC * NS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T *bt = reinterpret_cast<T *>(smem);
  T *cc = bt + LT * TC;
  T *xx = cc + LT * TC;
  T *zz = xx + LT * TC;
  T *gg = zz + LT * TC;
  float *dl = reinterpret_cast<float *>(gg + LT * TC);
  float *ddl = dl + LT * d.HT;
  float *segs = ddl + LT * d.HT;                                   // [NS][TC][3]
  float2 *lk = reinterpret_cast<float2 *>(segs + NS * TC * 3);     // [NS][TC]
  int *item_s = reinterpret_cast<int *>(lk + NS * TC);

  const int tid = threadIdx.x, lane = tid & 63, seg = tid >> 6;
  int chunk, ct, b;
  if constexpr (MODE == 1) {
    if (tid == 0) {
      const unsigned t = atomicAdd(&head->ctr[epoch & 1], 1u);
      if (t == 0) atomicExch(&head->ctr[(epoch + 1) & 1], 0u);
      *item_s = (int)t;
    }
    __syncthreads();
    const int item = *item_s, bct_n = (int)d.B * ctiles;
    const int k = item / bct_n, bct = item - k * bct_n;
    chunk = d.nchunks - 1 - k;                                     // right to left
    b = bct / ctiles;
    ct = bct - b * ctiles;
  } else {
    chunk = blockIdx.x; ct = blockIdx.y; b = blockIdx.z;
  }
  const int c0 = ct * TC, c = c0 + lane;
  const int64_t t0 = (int64_t)chunk * LT;
  const int rows_valid = (int)min((int64_t)LT, d.L - t0);
  const int ch_valid = (int)min((int64_t)TC, d.Dn - c0);
  const int ch_store = (int)min((int64_t)TC, store_w - c0);        // dBt / dC are zero-extended to the padded slice width
  const int64_t tok0 = (int64_t)b * d.L + t0;
  const int head0 = c0 >> d.log2N;
  const bool chan_ok = c < d.Dn;

  const int vb = ch_valid * (int)sizeof(T);
  stage_in<VB, ROWB, LT, NTH>(reinterpret_cast<char *>(cc), reinterpret_cast<const char *>(C + tok0 * c_rs + c0),
                              c_rs * sizeof(T), rows_valid, vb, tid);
  stage_in<VB, ROWB, LT, NTH>(reinterpret_cast<char *>(zz), reinterpret_cast<const char *>(z + tok0 * z_rs + c0),
                              z_rs * sizeof(T), rows_valid, vb, tid);
  stage_in<VB, ROWB, LT, NTH>(reinterpret_cast<char *>(gg), reinterpret_cast<const char *>(dout + tok0 * do_rs + c0),
                              do_rs * sizeof(T), rows_valid, vb, tid);
  stage_delta<LT, NTH>(dl, dlt, tok0, rows_valid, head0, (int)d.h, d.HT, d.softplus, tid);
  const float Ac = chan_ok ? -expf(A_log[c]) : 0.f;
  const float A2 = Ac * LOG2E_F;
  const float Dc = chan_ok ? Dv[c] : 0.f;
  const int64_t cidx = ((int64_t)b * d.nchunks + chunk) * d.Dn + c;
  float hcar = chan_ok ? h_in[cidx] : 0.f;
  TileRegs<VB, ROWB, LT, NTH> rb, rx;
  rb.load(reinterpret_cast<const char *>(Bt + tok0 * bt_rs + c0), bt_rs * sizeof(T), rows_valid, vb, tid);
  rx.load(reinterpret_cast<const char *>(xc + tok0 * xc_rs + c0), xc_rs * sizeof(T), rows_valid, vb, tid);
  __syncthreads();   // C, z, dout, delta tiles are in LDS

  float a[TS];
  float P = 1.f, M = 0.f;
  const int hh = lane >> d.log2N;
#pragma unroll
  for (int i = TS - 1; i >= 0; --i) {
    const int t = seg * TS + i;
    a[i] = __builtin_amdgcn_exp2f(dl[t * d.HT + hh] * A2);
    const float dv = to_f32(gg[t * TC + lane]) * silu_g(to_f32(zz[t * TC + lane]));
    const float u = dv * to_f32(cc[t * TC + lane]);
    M = a[i] * (u + M);
    P *= a[i];
  }
  segs[(seg * TC + lane) * 3 + 0] = P;
  segs[(seg * TC + lane) * 3 + 2] = M;
  __syncthreads();

  // mu entering from the right: the later chunks' reverse aggregates composed right to left
  {
    const int lo = chunk + 1, hi = d.nchunks, n = hi - lo, q = (n + NS - 1) / NS;
    const int s1 = max(hi - seg * q, lo), s0 = max(s1 - q, lo);   // wave `seg` takes the seg-th sub-range from the right
    float Pw = 1.f, Mw = 0.f;
    if constexpr (MODE == 1) {
      gran_t *gbase = gran + ((int64_t)(b * ctiles + ct) * d.nchunks) * TC * 2;
      if (seg == 0 && chunk > 0) {
        float Pt = segs[((NS - 1) * TC + lane) * 3 + 0], Mt = segs[((NS - 1) * TC + lane) * 3 + 2];
#pragma unroll
        for (int s = NS - 2; s >= 0; --s) {
          const float px = segs[(s * TC + lane) * 3 + 0], mx = segs[(s * TC + lane) * 3 + 2];
          Mt = fmaf(px, Mt, mx);
          Pt *= px;
        }
        if (chan_ok) {
          gran_store(gbase + ((int64_t)chunk * TC + lane) * 2 + 0, epoch, Pt);
          gran_store(gbase + ((int64_t)chunk * TC + lane) * 2 + 1, epoch, Mt);
        }
      }
      gather_published(gbase, s0, s1, true, lane, chan_ok, epoch, &head->err, Pw, Mw);
    } else {
      if (chan_ok) {
        const int64_t base = (int64_t)b * d.nchunks * d.Dn + c;
        for (int j = s1 - 1; j >= s0; --j) { const float2 r = agg[base + (int64_t)j * d.Dn]; Mw = fmaf(r.x, Mw, r.y); Pw *= r.x; }
      }
    }
    lk[seg * TC + lane] = make_float2(Pw, Mw);
  }
  rb.store(reinterpret_cast<char *>(bt), tid);
  rx.store(reinterpret_cast<char *>(xx), tid);
  __syncthreads();

  // forward segment aggregate (needs Bt) for the states inside the chunk
  float S = 0.f;
#pragma unroll
  for (int i = 0; i < TS; ++i) S = fmaf(a[i], S, to_f32(bt[(seg * TS + i) * TC + lane]));
  segs[(seg * TC + lane) * 3 + 1] = S;
  float mcar = 0.f;
#pragma unroll
  for (int s = 0; s < NS; ++s) { const float2 r = lk[s * TC + lane]; mcar = fmaf(r.x, mcar, r.y); }
  __syncthreads();
  for (int s = 0; s < seg; ++s) hcar = fmaf(segs[(s * TC + lane) * 3 + 0], hcar, segs[(s * TC + lane) * 3 + 1]);
  for (int s = NS - 1; s > seg; --s) mcar = fmaf(segs[(s * TC + lane) * 3 + 0], mcar, segs[(s * TC + lane) * 3 + 2]);

  float hs[TS];
  float hst = hcar;
#pragma unroll
  for (int i = 0; i < TS; ++i) {
    hst = fmaf(a[i], hst, to_f32(bt[(seg * TS + i) * TC + lane]));
    hs[i] = hst;
  }
  float mu = mcar, dA_acc = 0.f, dD_acc = 0.f;
#pragma unroll
  for (int i = TS - 1; i >= 0; --i) {
    const int t = seg * TS + i;
    const float g = to_f32(gg[t * TC + lane]), zf = to_f32(zz[t * TC + lane]);
    const float Cv = to_f32(cc[t * TC + lane]), xv = to_f32(xx[t * TC + lane]);
    const float dv = g * silu_g(zf);                       // d/dy = d/d(y + D*xc)
    const float lam = fmaf(dv, Cv, mu);
    const float hprev = i > 0 ? hs[i - 1] : hcar;
    const float q = lam * hprev * a[i] * Ac;               // da_t * a_t * A
    const float dlv = dl[t * d.HT + hh];
    dA_acc = fmaf(q, dlv, dA_acc);
    const float qs = group_sum(q, (int)d.N);
    if ((lane & ((int)d.N - 1)) == 0) ddl[t * d.HT + hh] = qs;
    const float yv = Cv * hs[i];
    const float dx = Dc * xv;
    const float v = yv + dx;
    dD_acc += dv * xv;
    cc[t * TC + lane] = from_f32<T>(dv * hs[i]);           // dC_t
    bt[t * TC + lane] = from_f32<T>(lam);                  // dBt_t
    zz[t * TC + lane] = from_f32<T>(g * v * silu_grad_g(zf));   // dz_t
    xx[t * TC + lane] = from_f32<T>(dv * Dc);              // dxc_t (the gate's share)
    mu = a[i] * lam;
  }
  __syncthreads();   // output tiles complete; segs / lk free
  segs[seg * TC + lane] = dA_acc;
  segs[(NS + seg) * TC + lane] = dD_acc;
  stage_out<VB, ROWB, LT, NTH>(reinterpret_cast<const char *>(bt), reinterpret_cast<char *>(dBt + tok0 * dbt_rs + c0),
                               dbt_rs * sizeof(T), rows_valid, ch_store * (int)sizeof(T), tid);
  stage_out<VB, ROWB, LT, NTH>(reinterpret_cast<const char *>(cc), reinterpret_cast<char *>(dC + tok0 * dc_rs + c0),
                               dc_rs * sizeof(T), rows_valid, ch_store * (int)sizeof(T), tid);
  stage_out<VB, ROWB, LT, NTH>(reinterpret_cast<const char *>(xx), reinterpret_cast<char *>(dxc + tok0 * dxc_rs + c0),
                               dxc_rs * sizeof(T), rows_valid, vb, tid);
  stage_out<VB, ROWB, LT, NTH>(reinterpret_cast<const char *>(zz), reinterpret_cast<char *>(dz + tok0 * dz_rs + c0),
                               dz_rs * sizeof(T), rows_valid, vb, tid);
  for (int idx = tid; idx < LT * d.HT; idx += NTH) {
    const int t = idx / d.HT, hx = idx - t * d.HT;
    if (t < rows_valid && head0 + hx < d.h) {
      float v = ddl[idx];
      if (d.softplus) v *= 1.f - expf(-dl[idx]);            // sigmoid(x) = 1 - exp(-softplus(x))
      d_dlt[(tok0 + t) * d.h + head0 + hx] = v;
    }
  }
  __syncthreads();
  if (seg < 2 && chan_ok) {                                  // wave 0: dA_log partial, wave 1: dD partial
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NS; ++w) s += segs[(seg * NS + w) * TC + lane];
    part[(((int64_t)b * d.nchunks + chunk) * 2 + seg) * d.Dn + c] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Single-token decode step (reference core.py:364-400 with L = 1 and a cache; generate() core.py:1578-1603):
//   window = [conv_state (k-1 tokens) | xp_new]; the reference keeps the FIRST conv output of that window
//   (core.py:369-373 slices [:, :, :L] of the padded conv), i.e. taps over [0, 0, 0, conv_state[0]] ... - reproduced
//   as is (front-slice quirk); new conv_state = last k-1 tokens of the window;
//   xc = silu(conv + bias)  ->  (caller: x_param_proj, dt)  ->  s = exp(softplus(dt)*A)*s + Bt;  out = (C*s + D*xc)*silu(z)
// Two kernels around the x_param_proj GEMM the caller runs: decode_conv_k (B x Dn threads) and decode_state_k.
template <typename T>
__global__ void __launch_bounds__(256)
decode_conv_k(const T *__restrict__ xp, int64_t xp_rs, const T *__restrict__ conv_state, T *__restrict__ conv_state_out,
              const float *__restrict__ w, const float *__restrict__ bias, T *__restrict__ xc, int64_t B, int64_t Dn, int k) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B * Dn) return;
  const int64_t b = i / Dn, c = i - b * Dn;
  // window of k tokens: conv_state[b, c, 0..k-2] then the new xp; conv1d(padding = k-1) output 0 sees inputs
  // [-(k-1) .. 0] = k-1 zeros and window[0], weighted by taps w[c, 0..k-1]: only the LAST tap meets a real value
  const T *cs = conv_state + (b * Dn + c) * (k - 1);
  const float first = k > 1 ? to_f32(cs[0]) : to_f32(xp[b * xp_rs + c]);
  const float acc = w[c * k + (k - 1)] * first + bias[c];
  xc[b * Dn + c] = from_f32<T>(acc / (1.f + expf(-acc)));
  // new cache = the last k-1 tokens of the window
  T *co = conv_state_out + (b * Dn + c) * (k - 1);
  for (int j = 0; j + 1 < k - 1; ++j) co[j] = cs[j + 1];
  if (k > 1) co[k - 2] = xp[b * xp_rs + c];
}

template <typename T>
__global__ void __launch_bounds__(256)
decode_state_k(const float *__restrict__ dt_logits, const float *__restrict__ A_log, const T *__restrict__ Bt, int64_t bt_rs,
               const T *__restrict__ C, int64_t c_rs, const T *__restrict__ xc, const T *__restrict__ z, int64_t z_rs,
               const float *__restrict__ Dv, float *__restrict__ state, T *__restrict__ out, int64_t B, int64_t h, int64_t N,
               int softplus) {
  const int64_t Dn = h * N;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B * Dn) return;
  const int64_t b = i / Dn, c = i - b * Dn;
  float dlv = dt_logits[b * h + c / N];
  if (softplus) dlv = softplus_f(dlv);
  const float av = __builtin_amdgcn_exp2f(dlv * (-expf(A_log[c]) * LOG2E_F));
  const float s = fmaf(av, state[b * Dn + c], to_f32(Bt[b * bt_rs + c]));
  state[b * Dn + c] = s;
  const float yv = to_f32(C[b * c_rs + c]) * s;
  const float dx = Dv[c] * to_f32(xc[b * Dn + c]);
  const float v = yv + dx;
  out[b * Dn + c] = from_f32<T>(v * silu_g(to_f32(z[b * z_rs + c])));
}

// ---------------------------------------------------------------------------------------------------------------
template <typename T> struct FwdGeo { static constexpr int LT = sizeof(T) == 2 ? 128 : 64, NS = 8; };
constexpr int LT_BWD = 64, NS_BWD = 8;

template <typename T> size_t fwd_lds(const ScanDims &d) {
  constexpr int LT = FwdGeo<T>::LT, NS = FwdGeo<T>::NS;
  return 4 * (size_t)LT * TC * sizeof(T) + (size_t)LT * d.HT * 4 + 2 * NS * TC * sizeof(float2) + 16;
}
template <typename T> size_t bwd_lds(const ScanDims &d) {
  return 5 * (size_t)LT_BWD * TC * sizeof(T) + 2 * (size_t)LT_BWD * d.HT * 4 + NS_BWD * TC * 3 * 4 + NS_BWD * TC * sizeof(float2) + 16;
}

template <typename F> void allow_lds(F *fn, size_t bytes) {
  if (bytes > 64 * 1024) (void)hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <typename T, int VB>
int launch_gate_fwd(const float *dlt, const float *A_log, const void *Bt, int64_t bt_rs, const void *C, int64_t c_rs,
                    const void *xc, int64_t xc_rs, const void *z, int64_t z_rs, const float *Dv, const float *h0, void *out,
                    int64_t out_rs, float *h_last, float *agg, float *h_in, void *ws, uint32_t epoch, const ScanDims &d64,
                    int single_pass, hipStream_t st) {
  constexpr int LT = FwdGeo<T>::LT, NS = FwdGeo<T>::NS;
  ScanDims d = d64;
  d.nchunks = (int)ceil_div64(d.L, LT);
  const int ctiles = (int)ceil_div64(d.Dn, TC);
  const size_t lds = fwd_lds<T>(d);
  GateWsHead *head = reinterpret_cast<GateWsHead *>(ws);
  gran_t *gran = reinterpret_cast<gran_t *>(reinterpret_cast<char *>(ws) + sizeof(GateWsHead));
  if (single_pass) {
    const int64_t items = (int64_t)d.nchunks * d.B * ctiles;
    if (items > 0x7fffffffLL) return APERTIS_ERR_UNSUPPORTED;
    allow_lds(scan_gate_fwd_k<T, VB, LT, NS, 1>, lds);
    hipLaunchKernelGGL((scan_gate_fwd_k<T, VB, LT, NS, 1>), dim3((unsigned)items), dim3(TC * NS), lds, st, dlt, A_log,
                       (const T *)Bt, bt_rs, (const T *)C, c_rs, (const T *)xc, xc_rs, (const T *)z, z_rs, Dv, h0,
                       (const float2 *)nullptr, head, gran, epoch, h_in, h_last, (T *)out, out_rs, d, (int64_t)d64.nchunks,
                       ctiles);
  } else {
    dim3 grid(d.nchunks, (unsigned)ctiles, (unsigned)d.B);
    const size_t lds1 = (size_t)LT * TC * sizeof(T) + (size_t)LT * d.HT * 4 + NS * TC * sizeof(float2);
    hipLaunchKernelGGL((scan_fwd_state<T, VB, LT, NS>), grid, dim3(TC * NS), lds1, st, dlt, A_log, (const T *)Bt, bt_rs,
                       (float2 *)agg, d);
    allow_lds(scan_gate_fwd_k<T, VB, LT, NS, 0>, lds);
    hipLaunchKernelGGL((scan_gate_fwd_k<T, VB, LT, NS, 0>), grid, dim3(TC * NS), lds, st, dlt, A_log, (const T *)Bt, bt_rs,
                       (const T *)C, c_rs, (const T *)xc, xc_rs, (const T *)z, z_rs, Dv, h0, (const float2 *)agg, head, gran,
                       epoch, h_in, h_last, (T *)out, out_rs, d, (int64_t)d64.nchunks, ctiles);
  }
  return apertis_check_launch();
}

template <typename T, int VB>
int launch_gate_bwd(const float *dlt, const float *A_log, const void *Bt, int64_t bt_rs, const void *C, int64_t c_rs,
                    const void *xc, int64_t xc_rs, const void *z, int64_t z_rs, const float *Dv, const void *dout,
                    int64_t do_rs, const float *h_in, void *dBt, int64_t dbt_rs, void *dC, int64_t dc_rs, int64_t store_w,
                    void *dxc, int64_t dxc_rs, void *dz, int64_t dz_rs, float *d_dlt, float *dA_dD, float *agg, float *fold,
                    float *part, void *ws, uint32_t epoch, const ScanDims &d, int single_pass, hipStream_t st) {
  constexpr int LT = LT_BWD, NS = NS_BWD;
  const int ctiles = (int)ceil_div64(d.Dn, TC);
  const size_t lds = bwd_lds<T>(d);
  GateWsHead *head = reinterpret_cast<GateWsHead *>(ws);
  gran_t *gran = reinterpret_cast<gran_t *>(reinterpret_cast<char *>(ws) + sizeof(GateWsHead));
  if (single_pass) {
    const int64_t items = (int64_t)d.nchunks * d.B * ctiles;
    if (items > 0x7fffffffLL) return APERTIS_ERR_UNSUPPORTED;
    allow_lds(scan_gate_bwd_k<T, VB, LT, NS, 1>, lds);
    hipLaunchKernelGGL((scan_gate_bwd_k<T, VB, LT, NS, 1>), dim3((unsigned)items), dim3(TC * NS), lds, st, dlt, A_log,
                       (const T *)Bt, bt_rs, (const T *)C, c_rs, (const T *)xc, xc_rs, (const T *)z, z_rs, Dv, (const T *)dout,
                       do_rs, h_in, (const float2 *)nullptr, head, gran, epoch, (T *)dBt, dbt_rs, (T *)dC, dc_rs, store_w,
                       (T *)dxc, dxc_rs, (T *)dz, dz_rs, d_dlt, part, d, ctiles);
  } else {
    dim3 grid(d.nchunks, (unsigned)ctiles, (unsigned)d.B);
    const size_t lds1 = 3 * (size_t)LT * TC * sizeof(T) + (size_t)LT * d.HT * 4 + NS * TC * sizeof(float2);
    hipLaunchKernelGGL((scan_gate_bwd_state_k<T, VB, LT, NS>), grid, dim3(TC * NS), lds1, st, dlt, A_log, (const T *)C, c_rs,
                       (const T *)z, z_rs, (const T *)dout, do_rs, (float2 *)agg, d);
    allow_lds(scan_gate_bwd_k<T, VB, LT, NS, 0>, lds);
    hipLaunchKernelGGL((scan_gate_bwd_k<T, VB, LT, NS, 0>), grid, dim3(TC * NS), lds, st, dlt, A_log, (const T *)Bt, bt_rs,
                       (const T *)C, c_rs, (const T *)xc, xc_rs, (const T *)z, z_rs, Dv, (const T *)dout, do_rs, h_in,
                       (const float2 *)agg, head, gran, epoch, (T *)dBt, dbt_rs, (T *)dC, dc_rs, store_w, (T *)dxc, dxc_rs,
                       (T *)dz, dz_rs, d_dlt, part, d, ctiles);
  }
  // fold the per-chunk partials [rows][2*Dn] (dA_log | dD) in a fixed order, two levels
  const int64_t rows = d.B * d.nchunks, cols = 2 * d.Dn;
  const unsigned ctl = (unsigned)ceil_div64(cols, TC);
  if (rows <= 128) {
    hipLaunchKernelGGL(colsum_kernel, dim3(ctl), dim3(1024), 0, st, part, dA_dD, rows, cols, rows);
  } else {
    const int64_t groups = std::min<int64_t>(64, ceil_div64(rows, 64)), rpg = ceil_div64(rows, groups);
    const int64_t ng = ceil_div64(rows, rpg);
    hipLaunchKernelGGL(colsum_kernel, dim3(ctl, (unsigned)ng), dim3(1024), 0, st, part, fold, rows, cols, rpg);
    hipLaunchKernelGGL(colsum_kernel, dim3(ctl), dim3(1024), 0, st, fold, dA_dD, ng, cols, ng);
  }
  return apertis_check_launch();
}

template <typename T> int gate_align(std::initializer_list<std::pair<const void *, int64_t>> slices, int64_t Dn) {
  int al = 16;
  for (auto &s : slices) al = std::min(al, slice_align<T>(s.first, s.second, Dn));
  return al;
}

}  // namespace

extern "C" int64_t apertis_scan_gate_workspace_bytes(int64_t B, int64_t L, int64_t Dn) {
  // 64-byte head (two ticket counters, error word) + [B * ctiles][chunks of 64 tokens][64 lanes][2 granules] x 8 bytes
  return (int64_t)sizeof(GateWsHead) + B * ceil_div64(Dn, TC) * ceil_div64(L, LT_DEFAULT) * TC * 2 * 8;
}

extern "C" int apertis_scan_gate_fwd(const float *dlt, const float *A_log, const void *Bt, int64_t bt_rs, const void *C,
                                     int64_t c_rs, const void *xc, int64_t xc_rs, const void *z, int64_t z_rs,
                                     const float *D, const float *h0, void *out, int64_t out_rs, float *h_last, float *agg,
                                     float *h_in, void *ws, uint32_t epoch, int64_t B, int64_t L, int64_t h, int64_t N,
                                     int dtype, int delta_softplus, int single_pass, void *stream) {
  if (!dlt || !A_log || !Bt || !C || !xc || !z || !D || !out || !h_in) return APERTIS_ERR_ARG;
  if (single_pass ? (!ws || epoch == 0) : !agg) return APERTIS_ERR_ARG;
  ScanDims d;
  int rc = make_dims(d, B, L, h, N, delta_softplus);
  if (rc) return rc;
  if (bt_rs < d.Dn || c_rs < d.Dn || xc_rs < d.Dn || z_rs < d.Dn || out_rs < d.Dn) return APERTIS_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
#define GF(T, VB) \
  return launch_gate_fwd<T, VB>(dlt, A_log, Bt, bt_rs, C, c_rs, xc, xc_rs, z, z_rs, D, h0, out, out_rs, h_last, agg, h_in, \
                                ws, epoch, d, single_pass, st)
  if (dtype == APERTIS_F32) {
    const int al = gate_align<float>({{Bt, bt_rs}, {C, c_rs}, {xc, xc_rs}, {z, z_rs}, {out, out_rs}}, d.Dn);
    if (al >= 16) GF(float, 16);
    if (al >= 8) GF(float, 8);
    GF(float, 4);
  } else if (dtype == APERTIS_BF16) {
    const int al = gate_align<bf16_t>({{Bt, bt_rs}, {C, c_rs}, {xc, xc_rs}, {z, z_rs}, {out, out_rs}}, d.Dn);
    if (al >= 16) GF(bf16_t, 16);
    if (al >= 8) GF(bf16_t, 8);
    if (al >= 4) GF(bf16_t, 4);
    GF(bf16_t, 2);
  }
#undef GF
  return APERTIS_ERR_UNSUPPORTED;
}

extern "C" int apertis_scan_gate_bwd(const float *dlt, const float *A_log, const void *Bt, int64_t bt_rs, const void *C,
                                     int64_t c_rs, const void *xc, int64_t xc_rs, const void *z, int64_t z_rs,
                                     const float *D, const void *dout, int64_t dout_rs, const float *h_in, void *dBt,
                                     int64_t dbt_rs, void *dC, int64_t dc_rs, int64_t store_w, void *dxc, int64_t dxc_rs,
                                     void *dz, int64_t dz_rs, float *d_dlt, float *dA_dD, float *agg, float *fold,
                                     float *part, void *ws, uint32_t epoch, int64_t B, int64_t L, int64_t h, int64_t N,
                                     int dtype, int delta_softplus, int single_pass, void *stream) {
  if (!dlt || !A_log || !Bt || !C || !xc || !z || !D || !dout || !h_in || !dBt || !dC || !dxc || !dz || !d_dlt || !dA_dD ||
      !fold || !part)
    return APERTIS_ERR_ARG;
  if (single_pass ? (!ws || epoch == 0) : !agg) return APERTIS_ERR_ARG;
  ScanDims d;
  int rc = make_dims(d, B, L, h, N, delta_softplus);
  if (rc) return rc;
  if (bt_rs < d.Dn || c_rs < d.Dn || xc_rs < d.Dn || z_rs < d.Dn || dout_rs < d.Dn || dxc_rs < d.Dn || dz_rs < d.Dn)
    return APERTIS_ERR_ARG;
  // dBt / dC may be zero-extended up to the next multiple of 64 channels (the padded slices of the projection output)
  if (store_w < d.Dn || store_w > ceil_div64(d.Dn, TC) * TC || dbt_rs < store_w || dc_rs < store_w) return APERTIS_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
#define GB(T, VB)                                                                                                      \
  return launch_gate_bwd<T, VB>(dlt, A_log, Bt, bt_rs, C, c_rs, xc, xc_rs, z, z_rs, D, dout, dout_rs, h_in, dBt, dbt_rs, dC, \
                                dc_rs, store_w, dxc, dxc_rs, dz, dz_rs, d_dlt, dA_dD, agg, fold, part, ws, epoch, d,       \
                                single_pass, st)
  if (dtype == APERTIS_F32) {
    int al = gate_align<float>({{Bt, bt_rs}, {C, c_rs}, {xc, xc_rs}, {z, z_rs}, {dout, dout_rs}, {dxc, dxc_rs}, {dz, dz_rs}}, d.Dn);
    al = std::min(al, gate_align<float>({{dBt, dbt_rs}, {dC, dc_rs}}, store_w));
    if (al >= 16) GB(float, 16);
    if (al >= 8) GB(float, 8);
    GB(float, 4);
  } else if (dtype == APERTIS_BF16) {
    int al = gate_align<bf16_t>({{Bt, bt_rs}, {C, c_rs}, {xc, xc_rs}, {z, z_rs}, {dout, dout_rs}, {dxc, dxc_rs}, {dz, dz_rs}}, d.Dn);
    al = std::min(al, gate_align<bf16_t>({{dBt, dbt_rs}, {dC, dc_rs}}, store_w));
    if (al >= 16) GB(bf16_t, 16);
    if (al >= 8) GB(bf16_t, 8);
    if (al >= 4) GB(bf16_t, 4);
    GB(bf16_t, 2);
  }
#undef GB
  return APERTIS_ERR_UNSUPPORTED;
}

extern "C" int apertis_ssm_decode_conv(const void *xp, int64_t xp_rs, const void *conv_state, void *conv_state_out,
                                       const float *w, const float *bias, void *xc, int64_t B, int64_t Dn, int64_t k,
                                       int dtype, void *stream) {
  if (!xp || !conv_state_out || !w || !bias || !xc || B <= 0 || Dn <= 0 || k < 1 || k > 16 || xp_rs < Dn) return APERTIS_ERR_ARG;
  if (k > 1 && !conv_state) return APERTIS_ERR_ARG;
  const unsigned grid = (unsigned)ceil_div64(B * Dn, 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == APERTIS_F32)
    hipLaunchKernelGGL(decode_conv_k<float>, dim3(grid), dim3(256), 0, st, (const float *)xp, xp_rs, (const float *)conv_state,
                       (float *)conv_state_out, w, bias, (float *)xc, B, Dn, (int)k);
  else if (dtype == APERTIS_BF16)
    hipLaunchKernelGGL(decode_conv_k<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t *)xp, xp_rs, (const bf16_t *)conv_state,
                       (bf16_t *)conv_state_out, w, bias, (bf16_t *)xc, B, Dn, (int)k);
  else
    return APERTIS_ERR_UNSUPPORTED;
  return apertis_check_launch();
}

extern "C" int apertis_ssm_decode_state(const float *dt_logits, const float *A_log, const void *Bt, int64_t bt_rs,
                                        const void *C, int64_t c_rs, const void *xc, const void *z, int64_t z_rs,
                                        const float *D, float *state, void *out, int64_t B, int64_t h, int64_t N, int dtype,
                                        int delta_softplus, void *stream) {
  if (!dt_logits || !A_log || !Bt || !C || !xc || !z || !D || !state || !out || B <= 0 || h <= 0 || N <= 0) return APERTIS_ERR_ARG;
  const int64_t Dn = h * N;
  if (bt_rs < Dn || c_rs < Dn || z_rs < Dn) return APERTIS_ERR_ARG;
  const unsigned grid = (unsigned)ceil_div64(B * Dn, 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == APERTIS_F32)
    hipLaunchKernelGGL(decode_state_k<float>, dim3(grid), dim3(256), 0, st, dt_logits, A_log, (const float *)Bt, bt_rs,
                       (const float *)C, c_rs, (const float *)xc, (const float *)z, z_rs, D, state, (float *)out, B, h, N,
                       delta_softplus);
  else if (dtype == APERTIS_BF16)
    hipLaunchKernelGGL(decode_state_k<bf16_t>, dim3(grid), dim3(256), 0, st, dt_logits, A_log, (const bf16_t *)Bt, bt_rs,
                       (const bf16_t *)C, c_rs, (const bf16_t *)xc, (const bf16_t *)z, z_rs, D, state, (bf16_t *)out, B, h, N,
                       delta_softplus);
  else
    return APERTIS_ERR_UNSUPPORTED;
  return apertis_check_launch();
}
