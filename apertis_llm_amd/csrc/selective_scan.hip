// Selective-SSM scan for gfx950 (MI355X): chunked associative scan, forward and backward.
//
// Reference semantics: SelectiveLinearAttention._ssm_pytorch_scan_recurrent
// (/root/reference/src/model/core.py:337-353).  The recurrence is element-wise on B*Dn
// independent channels:   s_t = a_t*s_{t-1} + Bt_t,   y_t = C_t*s_t,   a_t = exp(delta_t*A).
// Pairs (a, b) compose associatively as (a2*a1, a2*b1 + b2); nothing ever divides by a
// cumulative product (the reference's parallel form core.py:331 does and overflows).
//
// Structure (HBM-bound; algorithmic bytes per token = 3*Dn*e + 4*h fwd, 5*Dn*e + 8*h bwd):
//   work-group = (batch b, chunk of LT tokens, tile of 64 channels), 256 threads = 4 waves;
//   lane = channel (64 consecutive channels = one contiguous row segment of the token-major
//   [B,L,Dn] tensors), wave = a segment of LT/4 consecutive tokens.
//   1. tiles of Bt / C / dy are staged HBM -> LDS with the widest loads the slice alignment
//      allows (16 B per lane when possible), so HBM sees whole row segments;
//   2. each thread scans its own column of the LDS tile sequentially in fp32 registers;
//   3. segments are stitched through a 4x64 LDS table, chunks through a [B,nchunks,Dn,2]
//      fp32 aggregate workspace: pass 1 writes every chunk's (prod a, state-from-zero), pass 2
//      composes its own carry-in from the other chunks' aggregates and replays (two launches).
//   The backward recomputes the states inside the chunk from the saved chunk carry-in and
//   runs the adjoint recurrence right-to-left with mu_t = a_t*lambda_t as the carried value.
#include "scan_common.h"

namespace {

// pass 2 (forward): compose the carry-in from the other chunks' aggregates, replay the chunk, write y
// The forward walks chunks of LT = 128 tokens on NS = 8 waves (a quarter of the look-back reads of 64-token chunks:
// half as many work-groups, each composing half as many aggregates - at 64 tokens the look-back moved as many bytes
// through L2 as the tile loads did); the backward keeps 64-token chunks (its fp32 dy tile would halve the work-groups
// per CU at 128), so the carry-in is saved at 64-token granularity: h_in[b][2*chunk + {0,1}] from waves 0 and NS/2.
template <typename TIN, typename TY, int VB, int LT, int NS>
__global__ void __launch_bounds__(TC * NS)
scan_fwd_replay(const float *__restrict__ dlt, const float *__restrict__ A_log,
                const TIN *__restrict__ Bt, int64_t bt_rs, const TIN *__restrict__ C, int64_t c_rs,
                const float2 *__restrict__ agg, const float *__restrict__ h0, float *__restrict__ h_in,
                float *__restrict__ h_last, TY *__restrict__ y, int64_t y_rs, ScanDims d, int64_t nch64) {
  constexpr int ROWB = TC * sizeof(TIN);
  constexpr int TS = LT / NS;
  constexpr int NTH = TC * NS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  TIN *bt = reinterpret_cast<TIN *>(smem);
  TIN *cc = reinterpret_cast<TIN *>(smem + LT * ROWB);
  float *dl = reinterpret_cast<float *>(smem + 2 * LT * ROWB);
  float2 *segs = reinterpret_cast<float2 *>(smem + 2 * LT * ROWB + LT * d.HT * 4);
  float2 *lk = segs;   // look-back table: aliases segs (barrier below)

  const int tid = threadIdx.x, lane = tid & 63, seg = tid >> 6;
  const int chunk = blockIdx.x, ct = blockIdx.y, b = blockIdx.z;
  const int c0 = ct * TC, c = c0 + lane;
  const int64_t t0 = (int64_t)chunk * LT;
  const int rows_valid = (int)min((int64_t)LT, d.L - t0);
  const int ch_valid = (int)min((int64_t)TC, d.Dn - c0);
  const int64_t tok0 = (int64_t)b * d.L + t0;

  stage_in<VB, ROWB, LT, NTH>(reinterpret_cast<char *>(bt),
                         reinterpret_cast<const char *>(Bt + tok0 * bt_rs + c0),
                         bt_rs * sizeof(TIN), rows_valid, ch_valid * (int)sizeof(TIN), tid);
  stage_in<VB, ROWB, LT, NTH>(reinterpret_cast<char *>(cc),
                         reinterpret_cast<const char *>(C + tok0 * c_rs + c0),
                         c_rs * sizeof(TIN), rows_valid, ch_valid * (int)sizeof(TIN), tid);
  stage_delta<LT, NTH>(dl, dlt, tok0, rows_valid, c0 >> d.log2N, (int)d.h, d.HT, d.softplus, tid);
  const bool chan_ok = c < d.Dn;
  const float A2 = chan_ok ? -expf(A_log[c]) * LOG2E_F : 0.f;
  // carry-in from the other chunks' aggregates (its barrier also covers the staged tiles)
  float hcar = chunk_carry<NS>(agg, h0, b, chunk, c, chan_ok, d, seg, lane, lk, false);
  __syncthreads();   // every wave has read lk before segs (same LDS) is written

  float a[TS];
  float P = 1.f, S = 0.f;
  const int hh = lane >> d.log2N;
#pragma unroll
  for (int i = 0; i < TS; ++i) {
    int t = seg * TS + i;
    a[i] = __builtin_amdgcn_exp2f(dl[t * d.HT + hh] * A2);
    S = fmaf(a[i], S, to_f32(bt[t * TC + lane]));
    P *= a[i];
  }
  segs[seg * TC + lane] = make_float2(P, S);
  __syncthreads();
  for (int s = 0; s < seg; ++s) {
    float2 q = segs[s * TC + lane];
    hcar = fmaf(q.x, hcar, q.y);
  }
  // state entering each 64-token half of the chunk, saved for the backward
  if ((seg == 0 || seg == NS / 2) && chan_ok) {
    const int64_t j64 = (int64_t)chunk * (LT / 64) + (seg ? 1 : 0);
    if (j64 < nch64) h_in[((int64_t)b * nch64 + j64) * d.Dn + c] = hcar;
  }
  float hst = hcar;
  if constexpr (sizeof(TY) == 4) {
    // one dword per lane, 256 B per wave-instruction: full-rate plain stores
    TY *yp = y + (tok0 + seg * TS) * y_rs + c;
#pragma unroll
    for (int i = 0; i < TS; ++i) {
      int t = seg * TS + i;
      hst = fmaf(a[i], hst, to_f32(bt[t * TC + lane]));
      float yv = to_f32(cc[t * TC + lane]) * hst;
      if (chan_ok && t < rows_valid) yp[(int64_t)i * y_rs] = from_f32<TY>(yv);
    }
  } else {
    // 2-byte outputs: write in place into the C tile (each thread owns its column), then
    // store whole row segments
    static_assert(sizeof(TY) == sizeof(TIN), "bf16 y requires bf16 Bt/C");
#pragma unroll
    for (int i = 0; i < TS; ++i) {
      int t = seg * TS + i;
      hst = fmaf(a[i], hst, to_f32(bt[t * TC + lane]));
      float yv = to_f32(cc[t * TC + lane]) * hst;
      reinterpret_cast<TY *>(cc)[t * TC + lane] = from_f32<TY>(yv);
    }
    __syncthreads();
    // y rows are contiguous [.., Dn]; alignment of the y slice may be lower than Bt's: use 2 B
    // granularity unless y_rs and c0 allow wider (decided by the host through VB of y == VB)
    stage_out<VB, ROWB, LT, NTH>(reinterpret_cast<const char *>(cc),
                            reinterpret_cast<char *>(y + tok0 * y_rs + c0), y_rs * sizeof(TY),
                            rows_valid, ch_valid * (int)sizeof(TY), tid);
  }
  // final state (padding tokens are identities, so the last segment's state is s_{L-1})
  if (h_last && chunk == d.nchunks - 1 && seg == NS - 1 && chan_ok) h_last[(int64_t)b * d.Dn + c] = hst;
}

// ---------------------------------------------------------------------------------------
// backward pass 1: reverse chunk aggregates (P = prod a, M = mu at chunk start from zero)
//   u_t = dy_t*C_t,  mu_t = a_t*(u_t + mu_{t+1})
template <typename TIN, typename TY, int VB, int VBY, int LT, int NS>
__global__ void __launch_bounds__(TC * NS)
scan_bwd_state(const float *__restrict__ dlt, const float *__restrict__ A_log,
               const TIN *__restrict__ C, int64_t c_rs, const TY *__restrict__ dy, int64_t dy_rs,
               float2 *__restrict__ agg, ScanDims d) {
  constexpr int ROWB = TC * sizeof(TIN);
  constexpr int ROWY = TC * sizeof(TY);
  constexpr int TS = LT / NS;
  constexpr int NTH = TC * NS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  TIN *cc = reinterpret_cast<TIN *>(smem);
  TY *gy = reinterpret_cast<TY *>(smem + LT * ROWB);
  float *dl = reinterpret_cast<float *>(smem + LT * ROWB + LT * ROWY);
  float2 *segs = reinterpret_cast<float2 *>(smem + LT * ROWB + LT * ROWY + LT * d.HT * 4);

  const int tid = threadIdx.x, lane = tid & 63, seg = tid >> 6;
  const int chunk = blockIdx.x, ct = blockIdx.y, b = blockIdx.z;
  const int c0 = ct * TC, c = c0 + lane;
  const int64_t t0 = (int64_t)chunk * LT;
  const int rows_valid = (int)min((int64_t)LT, d.L - t0);
  const int ch_valid = (int)min((int64_t)TC, d.Dn - c0);
  const int64_t tok0 = (int64_t)b * d.L + t0;

  stage_in<VB, ROWB, LT, NTH>(reinterpret_cast<char *>(cc),
                         reinterpret_cast<const char *>(C + tok0 * c_rs + c0),
                         c_rs * sizeof(TIN), rows_valid, ch_valid * (int)sizeof(TIN), tid);
  stage_in<VBY, ROWY, LT, NTH>(reinterpret_cast<char *>(gy),
                          reinterpret_cast<const char *>(dy + tok0 * dy_rs + c0),
                          dy_rs * sizeof(TY), rows_valid, ch_valid * (int)sizeof(TY), tid);
  stage_delta<LT, NTH>(dl, dlt, tok0, rows_valid, c0 >> d.log2N, (int)d.h, d.HT, d.softplus, tid);
  const float A2 = c < d.Dn ? -expf(A_log[c]) * LOG2E_F : 0.f;
  __syncthreads();

  float P = 1.f, M = 0.f;
  const int hh = lane >> d.log2N;
#pragma unroll
  for (int i = TS - 1; i >= 0; --i) {
    int t = seg * TS + i;
    float a = __builtin_amdgcn_exp2f(dl[t * d.HT + hh] * A2);
    float u = to_f32(gy[t * TC + lane]) * to_f32(cc[t * TC + lane]);
    M = a * (u + M);
    P *= a;
  }
  segs[seg * TC + lane] = make_float2(P, M);
  __syncthreads();
  if (seg == 0 && c < d.Dn) {
    // compose right-to-left: start from the last segment
    float2 q = segs[(NS - 1) * TC + lane];
    float Pt = q.x, Mt = q.y;
#pragma unroll
    for (int s = NS - 2; s >= 0; --s) {
      float2 r = segs[s * TC + lane];
      Mt = fmaf(r.x, Mt, r.y);
      Pt *= r.x;
    }
    agg[((int64_t)b * d.nchunks + chunk) * d.Dn + c] = make_float2(Pt, Mt);
  }
}

// backward pass 3: recompute states in the chunk, run the adjoint right-to-left
// NS waves per work-group: 8 (segments of 8 tokens) halves the per-thread register arrays, so twice as many waves
// fit next to the same LDS tiles (the tile of fp32 dy makes LDS, not registers, the limit on work-groups per CU)
template <typename TIN, typename TY, int VB, int VBY, int LT, int NS>
__global__ void __launch_bounds__(TC * NS)
scan_bwd_replay(const float *__restrict__ dlt, const float *__restrict__ A_log,
                const TIN *__restrict__ Bt, int64_t bt_rs, const TIN *__restrict__ C, int64_t c_rs,
                const TY *__restrict__ dy, int64_t dy_rs, const float *__restrict__ h_in,
                const float2 *__restrict__ agg, TIN *__restrict__ dBt, int64_t dbt_rs,
                TIN *__restrict__ dC, int64_t dc_rs, float *__restrict__ d_dlt,
                float *__restrict__ dA_part, ScanDims d) {
  constexpr int ROWB = TC * sizeof(TIN);
  constexpr int ROWY = TC * sizeof(TY);
  constexpr int TS = LT / NS;
  constexpr int NTH = TC * NS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  TIN *bt = reinterpret_cast<TIN *>(smem);
  TIN *cc = reinterpret_cast<TIN *>(smem + LT * ROWB);
  TY *gy = reinterpret_cast<TY *>(smem + 2 * LT * ROWB);
  float *dl = reinterpret_cast<float *>(smem + 2 * LT * ROWB + LT * ROWY);
  float *ddl = dl + LT * d.HT;
  float *segs = ddl + LT * d.HT;  // [NS][TC][3]
  float2 *lk = reinterpret_cast<float2 *>(segs);   // look-back table: aliases segs (barrier below), 40 KiB in all = 4 per CU

  const int tid = threadIdx.x, lane = tid & 63, seg = tid >> 6;
  const int chunk = blockIdx.x, ct = blockIdx.y, b = blockIdx.z;
  const int c0 = ct * TC, c = c0 + lane;
  const int64_t t0 = (int64_t)chunk * LT;
  const int rows_valid = (int)min((int64_t)LT, d.L - t0);
  const int ch_valid = (int)min((int64_t)TC, d.Dn - c0);
  const int64_t tok0 = (int64_t)b * d.L + t0;
  const int head0 = c0 >> d.log2N;

  stage_in<VB, ROWB, LT, NTH>(reinterpret_cast<char *>(bt),
                         reinterpret_cast<const char *>(Bt + tok0 * bt_rs + c0),
                         bt_rs * sizeof(TIN), rows_valid, ch_valid * (int)sizeof(TIN), tid);
  stage_in<VB, ROWB, LT, NTH>(reinterpret_cast<char *>(cc),
                         reinterpret_cast<const char *>(C + tok0 * c_rs + c0),
                         c_rs * sizeof(TIN), rows_valid, ch_valid * (int)sizeof(TIN), tid);
  stage_in<VBY, ROWY, LT, NTH>(reinterpret_cast<char *>(gy),
                          reinterpret_cast<const char *>(dy + tok0 * dy_rs + c0),
                          dy_rs * sizeof(TY), rows_valid, ch_valid * (int)sizeof(TY), tid);
  stage_delta<LT, NTH>(dl, dlt, tok0, rows_valid, head0, (int)d.h, d.HT, d.softplus, tid);
  const bool chan_ok = c < d.Dn;
  const float Ac = chan_ok ? -expf(A_log[c]) : 0.f;
  const float A2 = Ac * LOG2E_F;
  const int64_t cidx = ((int64_t)b * d.nchunks + chunk) * d.Dn + c;
  float hcar = chan_ok ? h_in[cidx] : 0.f;
  // mu entering from the right: composed from the later chunks' reverse aggregates
  float mcar = chunk_carry<NS>(agg, nullptr, b, chunk, c, chan_ok, d, seg, lane, lk, true);
  __syncthreads();   // every wave has read lk before segs (same LDS) is written

  float a[TS], hs[TS];
  float P = 1.f, S = 0.f, M = 0.f;
  const int hh = lane >> d.log2N;
#pragma unroll
  for (int i = 0; i < TS; ++i) {
    int t = seg * TS + i;
    a[i] = __builtin_amdgcn_exp2f(dl[t * d.HT + hh] * A2);
    S = fmaf(a[i], S, to_f32(bt[t * TC + lane]));
    P *= a[i];
  }
#pragma unroll
  for (int i = TS - 1; i >= 0; --i) {
    int t = seg * TS + i;
    float u = to_f32(gy[t * TC + lane]) * to_f32(cc[t * TC + lane]);
    M = a[i] * (u + M);
  }
  segs[(seg * TC + lane) * 3 + 0] = P;
  segs[(seg * TC + lane) * 3 + 1] = S;
  segs[(seg * TC + lane) * 3 + 2] = M;
  __syncthreads();
  for (int s = 0; s < seg; ++s)
    hcar = fmaf(segs[(s * TC + lane) * 3 + 0], hcar, segs[(s * TC + lane) * 3 + 1]);
  for (int s = NS - 1; s > seg; --s)
    mcar = fmaf(segs[(s * TC + lane) * 3 + 0], mcar, segs[(s * TC + lane) * 3 + 2]);

  // forward recompute of the states of this segment
  float hst = hcar;
#pragma unroll
  for (int i = 0; i < TS; ++i) {
    int t = seg * TS + i;
    hst = fmaf(a[i], hst, to_f32(bt[t * TC + lane]));
    hs[i] = hst;
  }
  // adjoint, right to left.  lambda_t = u_t + mu_{t+1};  mu_t = a_t*lambda_t
  float mu = mcar, dA_acc = 0.f;
#pragma unroll
  for (int i = TS - 1; i >= 0; --i) {
    int t = seg * TS + i;
    float g = to_f32(gy[t * TC + lane]);
    float lam = fmaf(g, to_f32(cc[t * TC + lane]), mu);
    float hprev = i > 0 ? hs[i - 1] : hcar;
    float q = lam * hprev * a[i] * Ac;  // dL/d(delta*A) * A  = da_t * a_t * A
    float dlv = dl[t * d.HT + hh];
    dA_acc = fmaf(q, dlv, dA_acc);       // dA_log[c] += da_t*a_t*delta_t*A
    float qs = group_sum(q, (int)d.N);   // d delta[t, head] = sum_n da_t*a_t*A
    if ((lane & ((int)d.N - 1)) == 0) ddl[t * d.HT + hh] = qs;
    cc[t * TC + lane] = from_f32<TIN>(g * hs[i]);  // dC_t (in place: own column only)
    bt[t * TC + lane] = from_f32<TIN>(lam);        // dBt_t
    mu = a[i] * lam;
  }
  __syncthreads();  // dBt/dC/ddl tiles complete; segs free for reuse
  segs[seg * TC + lane] = dA_acc;
  stage_out<VB, ROWB, LT, NTH>(reinterpret_cast<const char *>(bt),
                          reinterpret_cast<char *>(dBt + tok0 * dbt_rs + c0), dbt_rs * sizeof(TIN),
                          rows_valid, ch_valid * (int)sizeof(TIN), tid);
  stage_out<VB, ROWB, LT, NTH>(reinterpret_cast<const char *>(cc),
                          reinterpret_cast<char *>(dC + tok0 * dc_rs + c0), dc_rs * sizeof(TIN),
                          rows_valid, ch_valid * (int)sizeof(TIN), tid);
  for (int idx = tid; idx < LT * d.HT; idx += NTH) {
    int t = idx / d.HT, hx = idx - t * d.HT;
    if (t < rows_valid && head0 + hx < d.h) {
      float v = ddl[idx];
      if (d.softplus) v *= 1.f - expf(-dl[idx]);  // sigmoid(x) = 1 - exp(-softplus(x))
      d_dlt[(tok0 + t) * d.h + head0 + hx] = v;
    }
  }
  __syncthreads();
  if (seg == 0 && chan_ok) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NS; ++w) s += segs[w * TC + lane];
    dA_part[cidx] = s;
  }
}

}  // namespace

extern "C" int64_t apertis_scan_chunk_len(int64_t, int64_t, int64_t) { return LT_DEFAULT; }
extern "C" int64_t apertis_scan_num_chunks(int64_t, int64_t L, int64_t) {
  return ceil_div64(L, LT_DEFAULT);
}

namespace {

template <typename TIN, typename TY, int VB>
int launch_fwd(const float *dlt, const float *A_log, const void *Bt, int64_t bt_rs, const void *C,
               int64_t c_rs, const float *h0, void *y, int64_t y_rs, float *h_last, float *agg,
               float *h_in, const ScanDims &d64, hipStream_t st) {
  constexpr int LT = LT_FWD, NS = NS_FWD;
  static_assert(LT == 2 * LT_DEFAULT && (NS / 2) * (LT / NS) == LT_DEFAULT, "h_in is exported at LT_DEFAULT granularity by waves 0 and NS/2");
  ScanDims d = d64;
  d.nchunks = (int)ceil_div64(d.L, LT);   // the forward's own chunking (agg needs no more than the ABI's workspace)
  dim3 grid(d.nchunks, (unsigned)ceil_div64(d.Dn, TC), (unsigned)d.B), block(TC * NS);
  size_t dlb = (size_t)LT * d.HT * 4;
  size_t lds1 = LT * TC * sizeof(TIN) + dlb + NS * TC * sizeof(float2);
  size_t lds3 = 2 * LT * TC * sizeof(TIN) + dlb + NS * TC * sizeof(float2);
  hipLaunchKernelGGL((scan_fwd_state<TIN, VB, LT, NS>), grid, block, lds1, st, dlt, A_log,
                     (const TIN *)Bt, bt_rs, (float2 *)agg, d);
  if (lds3 > 64 * 1024)   // fp32 tiles: 70 KiB
    hipFuncSetAttribute((const void *)scan_fwd_replay<TIN, TY, VB, LT, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
  hipLaunchKernelGGL((scan_fwd_replay<TIN, TY, VB, LT, NS>), grid, block, lds3, st, dlt, A_log,
                     (const TIN *)Bt, bt_rs, (const TIN *)C, c_rs, (const float2 *)agg, h0, h_in, h_last, (TY *)y,
                     y_rs, d, (int64_t)d64.nchunks);
  return apertis_check_launch();
}

template <typename TIN, typename TY, int VB, int VBY>
int launch_bwd(const float *dlt, const float *A_log, const void *Bt, int64_t bt_rs, const void *C,
               int64_t c_rs, const void *dy, int64_t dy_rs, const float *h_in, void *dBt,
               int64_t dbt_rs, void *dC, int64_t dc_rs, float *d_dlt, float *dA_log, float *agg,
               float *mu_in, float *dA_part, const ScanDims &d, hipStream_t st) {
  // mu_in: [B, nchunks, Dn] fp32 workspace of the ABI (the reverse carry is composed in-kernel since r1); its head holds
  // the row-group sums of the two-level dA_log fold
  constexpr int LT = LT_DEFAULT;
  dim3 grid(d.nchunks, (unsigned)ceil_div64(d.Dn, TC), (unsigned)d.B), block(NTHREADS);
  size_t dlb = (size_t)LT * d.HT * 4;
  constexpr int NS3 = 8;   // waves per work-group of both streaming passes
  size_t lds1 = LT * TC * (sizeof(TIN) + sizeof(TY)) + dlb + NS3 * TC * sizeof(float2);
  size_t lds3 = LT * TC * (2 * sizeof(TIN) + sizeof(TY)) + 2 * dlb + NS3 * TC * 3 * sizeof(float);
  hipLaunchKernelGGL((scan_bwd_state<TIN, TY, VB, VBY, LT, NS3>), grid, dim3(TC * NS3), lds1, st, dlt, A_log,
                     (const TIN *)C, c_rs, (const TY *)dy, dy_rs, (float2 *)agg, d);
  hipLaunchKernelGGL((scan_bwd_replay<TIN, TY, VB, VBY, LT, NS3>), grid, dim3(TC * NS3), lds3, st, dlt, A_log,
                     (const TIN *)Bt, bt_rs, (const TIN *)C, c_rs, (const TY *)dy, dy_rs, h_in,
                     (const float2 *)agg, (TIN *)dBt, dbt_rs, (TIN *)dC, dc_rs, d_dlt, dA_part, d);
  const int64_t rows = d.B * d.nchunks;
  const unsigned ctiles = (unsigned)ceil_div64(d.Dn, TC);
  if (rows <= 128) {
    hipLaunchKernelGGL(colsum_kernel, dim3(ctiles), dim3(1024), 0, st, dA_part, dA_log, rows, d.Dn, rows);
  } else {
    const int64_t groups = std::min<int64_t>(64, ceil_div64(rows, 64)), rpg = ceil_div64(rows, groups);
    const int64_t ng = ceil_div64(rows, rpg);   // <= groups <= rows: fits the head of mu_in
    hipLaunchKernelGGL(colsum_kernel, dim3(ctiles, (unsigned)ng), dim3(1024), 0, st, dA_part, mu_in, rows, d.Dn, rpg);
    hipLaunchKernelGGL(colsum_kernel, dim3(ctiles), dim3(1024), 0, st, mu_in, dA_log, ng, d.Dn, ng);
  }
  return apertis_check_launch();
}

}  // namespace

extern "C" int apertis_selective_scan_fwd(const float *dlt, const float *A_log, const void *Bt,
                                          int64_t bt_rs, const void *C, int64_t c_rs,
                                          const float *h0, void *y, int64_t y_rs, float *h_last,
                                          float *agg, float *h_in, int64_t B, int64_t L, int64_t h,
                                          int64_t N, int dtype_bc, int dtype_y, int delta_softplus,
                                          void *stream) {
  if (!dlt || !A_log || !Bt || !C || !y || !agg || !h_in) return APERTIS_ERR_ARG;
  ScanDims d;
  int rc = make_dims(d, B, L, h, N, delta_softplus);
  if (rc) return rc;
  if (bt_rs < d.Dn || c_rs < d.Dn || y_rs < d.Dn) return APERTIS_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
#define FWD(TIN, TY, VB) \
  return launch_fwd<TIN, TY, VB>(dlt, A_log, Bt, bt_rs, C, c_rs, h0, y, y_rs, h_last, agg, h_in, d, st)
  if (dtype_bc == APERTIS_F32 && dtype_y == APERTIS_F32) {
    int al = std::min(slice_align<float>(Bt, bt_rs, d.Dn), slice_align<float>(C, c_rs, d.Dn));
    if (al >= 16) FWD(float, float, 16);
    if (al >= 8) FWD(float, float, 8);
    FWD(float, float, 4);
  } else if (dtype_bc == APERTIS_BF16 && dtype_y == APERTIS_F32) {
    int al = std::min(slice_align<bf16_t>(Bt, bt_rs, d.Dn), slice_align<bf16_t>(C, c_rs, d.Dn));
    if (al >= 16) FWD(bf16_t, float, 16);
    if (al >= 8) FWD(bf16_t, float, 8);
    if (al >= 4) FWD(bf16_t, float, 4);
    FWD(bf16_t, float, 2);
  } else if (dtype_bc == APERTIS_BF16 && dtype_y == APERTIS_BF16) {
    int al = std::min({slice_align<bf16_t>(Bt, bt_rs, d.Dn), slice_align<bf16_t>(C, c_rs, d.Dn),
                       slice_align<bf16_t>(y, y_rs, d.Dn)});
    if (al >= 16) FWD(bf16_t, bf16_t, 16);
    if (al >= 8) FWD(bf16_t, bf16_t, 8);
    if (al >= 4) FWD(bf16_t, bf16_t, 4);
    FWD(bf16_t, bf16_t, 2);
  }
#undef FWD
  return APERTIS_ERR_UNSUPPORTED;
}

extern "C" int apertis_selective_scan_bwd(const float *dlt, const float *A_log, const void *Bt,
                                          int64_t bt_rs, const void *C, int64_t c_rs,
                                          const void *dy, int64_t dy_rs, const float *h_in,
                                          void *dBt, int64_t dbt_rs, void *dC, int64_t dc_rs,
                                          float *d_dlt, float *dA_log, float *agg, float *mu_in,
                                          float *dA_part, int64_t B, int64_t L, int64_t h,
                                          int64_t N, int dtype_bc, int dtype_y, int delta_softplus,
                                          void *stream) {
  if (!dlt || !A_log || !Bt || !C || !dy || !h_in || !dBt || !dC || !d_dlt || !dA_log || !agg ||
      !mu_in || !dA_part)
    return APERTIS_ERR_ARG;
  ScanDims d;
  int rc = make_dims(d, B, L, h, N, delta_softplus);
  if (rc) return rc;
  if (bt_rs < d.Dn || c_rs < d.Dn || dy_rs < d.Dn || dbt_rs < d.Dn || dc_rs < d.Dn)
    return APERTIS_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
#define BWD(TIN, TY, VB, VBY)                                                                   \
  return launch_bwd<TIN, TY, VB, VBY>(dlt, A_log, Bt, bt_rs, C, c_rs, dy, dy_rs, h_in, dBt,     \
                                      dbt_rs, dC, dc_rs, d_dlt, dA_log, agg, mu_in, dA_part, d, st)
  if (dtype_bc == APERTIS_F32 && dtype_y == APERTIS_F32) {
    int al = std::min({slice_align<float>(Bt, bt_rs, d.Dn), slice_align<float>(C, c_rs, d.Dn),
                       slice_align<float>(dBt, dbt_rs, d.Dn), slice_align<float>(dC, dc_rs, d.Dn),
                       slice_align<float>(dy, dy_rs, d.Dn)});
    if (al >= 16) BWD(float, float, 16, 16);
    if (al >= 8) BWD(float, float, 8, 8);
    BWD(float, float, 4, 4);
  } else if (dtype_bc == APERTIS_BF16) {
    int al = std::min({slice_align<bf16_t>(Bt, bt_rs, d.Dn), slice_align<bf16_t>(C, c_rs, d.Dn),
                       slice_align<bf16_t>(dBt, dbt_rs, d.Dn), slice_align<bf16_t>(dC, dc_rs, d.Dn)});
    if (dtype_y == APERTIS_F32) {
      int aly = slice_align<float>(dy, dy_rs, d.Dn);
      if (al >= 16 && aly >= 16) BWD(bf16_t, float, 16, 16);
      if (al >= 8 && aly >= 16) BWD(bf16_t, float, 8, 16);
      if (al >= 8 && aly >= 8) BWD(bf16_t, float, 8, 8);
      if (al >= 4) BWD(bf16_t, float, 4, 4);
      BWD(bf16_t, float, 2, 4);
    } else if (dtype_y == APERTIS_BF16) {
      int aly = slice_align<bf16_t>(dy, dy_rs, d.Dn);
      int a2 = std::min(al, aly);
      if (a2 >= 16) BWD(bf16_t, bf16_t, 16, 16);
      if (a2 >= 8) BWD(bf16_t, bf16_t, 8, 8);
      if (a2 >= 4) BWD(bf16_t, bf16_t, 4, 4);
      BWD(bf16_t, bf16_t, 2, 2);
    }
  }
#undef BWD
  return APERTIS_ERR_UNSUPPORTED;
}
