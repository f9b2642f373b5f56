// Shared pieces of the selective-scan kernels (selective_scan.hip: plain scan; scan_gate.hip: scan with the skip + gate
// epilogue fused): tile staging HBM <-> LDS, the delta tile, chunk-carry composition, the fixed-order column sum and the
// host-side shape / alignment helpers.  Everything lives in an anonymous namespace: each translation unit gets its own copy.
#pragma once
#include "common.h"

namespace {

constexpr int TC = 64;     // channels per tile (= wave width)
constexpr int NSEG = 4;    // token segments per tile (= waves per work-group)
constexpr int NTHREADS = TC * NSEG;

template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// sum over an aligned group of `n` lanes (n power of two <= 64); every lane gets the sum
__device__ __forceinline__ float group_sum(float v, int n) {
  if (n == 16) {  // one DPP row: rotate-and-add, no LDS traffic
    v += dpp_mov<0x128>(v);  // row_ror:8
    v += dpp_mov<0x124>(v);  // row_ror:4
    v += dpp_mov<0x122>(v);  // row_ror:2
    v += dpp_mov<0x121>(v);  // row_ror:1
    return v;
  }
  for (int off = n >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__device__ __forceinline__ float softplus_f(float x) {
  // torch.nn.functional.softplus(beta=1, threshold=20)
  return x > 20.f ? x : log1pf(expf(x));
}

template <int VB> struct vec_bytes;
template <> struct vec_bytes<16> { typedef uint4 type; };
template <> struct vec_bytes<8> { typedef uint2 type; };
template <> struct vec_bytes<4> { typedef uint32_t type; };
template <> struct vec_bytes<2> { typedef uint16_t type; };

// streaming output store of VB bytes: non-temporal (these outputs are larger than what the caches can hand to their consumer;
// written through them they displace what the next kernels read - measured on the whole step)
template <int VB> __device__ __forceinline__ void nt_store(void *p, const typename vec_bytes<VB>::type &v) {
  if constexpr (VB == 16) {
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    __builtin_nontemporal_store(__builtin_bit_cast(u4, v), reinterpret_cast<u4 *>(p));
  } else if constexpr (VB == 8) {
    typedef __attribute__((ext_vector_type(2))) unsigned u2;
    __builtin_nontemporal_store(__builtin_bit_cast(u2, v), reinterpret_cast<u2 *>(p));
  } else {
    __builtin_nontemporal_store(v, reinterpret_cast<typename vec_bytes<VB>::type *>(p));
  }
}

// streaming input load of VB bytes (read once by this kernel): non-temporal
template <int VB> __device__ __forceinline__ typename vec_bytes<VB>::type nt_load(const void *p) {
  typedef typename vec_bytes<VB>::type V;
  if constexpr (VB == 16) {
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    return __builtin_bit_cast(V, __builtin_nontemporal_load(reinterpret_cast<const u4 *>(p)));
  } else if constexpr (VB == 8) {
    typedef __attribute__((ext_vector_type(2))) unsigned u2;
    return __builtin_bit_cast(V, __builtin_nontemporal_load(reinterpret_cast<const u2 *>(p)));
  } else {
    return __builtin_nontemporal_load(reinterpret_cast<const V *>(p));
  }
}

template <int VB> __device__ __forceinline__ typename vec_bytes<VB>::type zero_vec() {
  typename vec_bytes<VB>::type z;
  __builtin_memset(&z, 0, VB);
  return z;
}

// HBM -> LDS: LT rows of ROWB bytes each (LDS pitch = ROWB), source rows `rsb` bytes apart.
// Rows >= rows_valid and bytes >= bytes_valid are zero-filled.
template <int VB, int ROWB, int LT, int NTH = NTHREADS>
__device__ __forceinline__ void stage_in(char *lds, const char *g, int64_t rsb, int rows_valid,
                                         int bytes_valid, int tid) {
  typedef typename vec_bytes<VB>::type V;
  constexpr int CPR = ROWB / VB;
  constexpr int TOTAL = LT * CPR;
  constexpr int ITERS = (TOTAL + NTH - 1) / NTH;
  V regs[ITERS];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    int idx = tid + it * NTH;
    int row = idx / CPR, cb = (idx % CPR) * VB;
    bool ok = idx < TOTAL && row < rows_valid && cb < bytes_valid;
    regs[it] = ok ? *reinterpret_cast<const V *>(g + (int64_t)row * rsb + cb) : zero_vec<VB>();
  }
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    int idx = tid + it * NTH;
    if (idx < TOTAL) *reinterpret_cast<V *>(lds + idx * VB) = regs[it];
  }
}

// LDS -> HBM, mirror of stage_in.
template <int VB, int ROWB, int LT, int NTH = NTHREADS>
__device__ __forceinline__ void stage_out(const char *lds, char *g, int64_t rsb, int rows_valid,
                                          int bytes_valid, int tid) {
  typedef typename vec_bytes<VB>::type V;
  constexpr int CPR = ROWB / VB;
  constexpr int TOTAL = LT * CPR;
  constexpr int ITERS = (TOTAL + NTH - 1) / NTH;
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    int idx = tid + it * NTH;
    int row = idx / CPR, cb = (idx % CPR) * VB;
    if (idx < TOTAL && row < rows_valid && cb < bytes_valid)
      *reinterpret_cast<V *>(g + (int64_t)row * rsb + cb) =
          *reinterpret_cast<const V *>(lds + idx * VB);
  }
}

// delta tile: dl[t][hh] for the HT = 64/N heads covered by the channel tile
template <int LT, int NTH = NTHREADS>
__device__ __forceinline__ void stage_delta(float *dl, const float *dlt, int64_t tok0, int rows_valid,
                                            int head0, int h, int HT, int softplus, int tid) {
  for (int idx = tid; idx < LT * HT; idx += NTH) {
    int t = idx / HT, hh = idx - t * HT;
    float v = 0.f;
    if (t < rows_valid && head0 + hh < h) {
      v = dlt[(tok0 + t) * h + head0 + hh];
      if (softplus) v = softplus_f(v);
    }
    dl[idx] = v;
  }
}

struct ScanDims {
  int64_t B, L, h, N, Dn;
  int log2N, HT, nchunks, softplus;
  int64_t nck;   // rows of the lean forward's checkpoint table per batch: ceil(L / 4)
};

// Carry entering chunk `chunk` for this lane's channel, composed from the aggregates of the
// other chunks (all final: pass 1 has completed).  The 4 waves split the range, partials meet in
// a 4x64 LDS table.  forward: chunks [0, chunk) left-to-right from `init`; reverse: chunks
// (chunk, nchunks) right-to-left from 0.  Replaces a separate prefix launch (which cost as much as
// the streaming passes at B*L = 32k tokens).  Contains one __syncthreads().
template <int NS = NSEG>
__device__ __forceinline__ float chunk_carry(const float2 *__restrict__ agg, const float *__restrict__ init, int b,
                                             int chunk, int c, bool chan_ok, const ScanDims &d, int seg, int lane,
                                             float2 *lk, bool reverse) {
  const int lo = reverse ? chunk + 1 : 0, hi = reverse ? d.nchunks : chunk;   // [lo, hi)
  const int n = hi - lo, q = (n + NS - 1) / NS;
  // wave `seg` takes the seg-th sub-range in COMPOSITION order
  int s0 = lo + seg * q, s1 = min(s0 + q, hi);
  if (reverse) { s1 = hi - seg * q; s0 = max(s1 - q, lo); }
  float P = 1.f, S = 0.f;
  if (chan_ok) {
    const int64_t base = (int64_t)b * d.nchunks * d.Dn + c;
    if (!reverse)
      for (int j = s0; j < s1; ++j) { float2 t = agg[base + (int64_t)j * d.Dn]; S = fmaf(t.x, S, t.y); P *= t.x; }
    else
      for (int j = s1 - 1; j >= s0; --j) { float2 t = agg[base + (int64_t)j * d.Dn]; S = fmaf(t.x, S, t.y); P *= t.x; }
  }
  lk[seg * TC + lane] = make_float2(P, S);
  __syncthreads();
  float carry = (init && chan_ok) ? init[(int64_t)b * d.Dn + c] : 0.f;
#pragma unroll
  for (int s = 0; s < NS; ++s) { float2 t = lk[s * TC + lane]; carry = fmaf(t.x, carry, t.y); }
  return carry;
}

// column sums of a [rows, cols] fp32 matrix in a fixed order: block (x, y) sums rows [y*rpg, (y+1)*rpg) of its 64
// columns into out[y][c].  Two levels (row groups, then the group sums): a single level leaves the whole matrix to
// cols/64 work-groups - 3 at Dn = 176, 15.6 us for 1.4 MB at the bench shape, a tenth of the backward.
__global__ void __launch_bounds__(1024)
colsum_kernel(const float *__restrict__ in, float *__restrict__ out, int64_t rows, int64_t cols, int64_t rpg) {
  __shared__ float part[16][TC];
  const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * TC + lane;
  const int64_t r0 = (int64_t)blockIdx.y * rpg, r1 = min(r0 + rpg, rows);
  float s = 0.f;
  if (c < cols) {
    int64_t r = r0 + seg;
    for (; r + 48 < r1; r += 64) {
      float a0 = in[r * cols + c], a1 = in[(r + 16) * cols + c], a2 = in[(r + 32) * cols + c], a3 = in[(r + 48) * cols + c];
      s += (a0 + a1) + (a2 + a3);
    }
    for (; r < r1; r += 16) s += in[r * cols + c];
  }
  part[seg][lane] = s;
  __syncthreads();
  if (seg == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += part[i][lane];
    out[(int64_t)blockIdx.y * cols + c] = t;
  }
}

int ilog2_exact(int64_t n) {
  int l = 0;
  while ((1LL << l) < n) ++l;
  return (1LL << l) == n ? l : -1;
}

constexpr int LT_DEFAULT = 64;   // chunk length of the ABI's workspaces (agg, h_in, dA_part) and of the backward
constexpr int LT_FWD = 128, NS_FWD = 8;   // the forward's own chunking

int make_dims(ScanDims &d, int64_t B, int64_t L, int64_t h, int64_t N, int softplus) {
  if (B <= 0 || L <= 0 || h <= 0 || N <= 0) return APERTIS_ERR_ARG;
  int l2 = ilog2_exact(N);
  if (l2 < 0 || N > TC) return APERTIS_ERR_UNSUPPORTED;  // d_state: power of two <= 64
  d.B = B; d.L = L; d.h = h; d.N = N; d.Dn = h * N;
  d.log2N = l2; d.HT = (int)(TC / N);
  d.nchunks = (int)ceil_div64(L, LT_DEFAULT);
  d.softplus = softplus;
  d.nck = ceil_div64(L, 4);
  if (B > 65535 || ceil_div64(d.Dn, TC) > 65535) return APERTIS_ERR_UNSUPPORTED;
  return APERTIS_OK;
}

template <typename T> int slice_align(const void *p, int64_t rs, int64_t Dn) {
  // bytes: pointer, row stride, the 64-channel tile step and the row length must all be
  // multiples of the access width (so no access straddles the end of a row slice)
  return common_align({(uint64_t)(uintptr_t)p, (uint64_t)rs * sizeof(T), (uint64_t)TC * sizeof(T),
                       (uint64_t)Dn * sizeof(T)});
}

// ---------------------------------------------------------------------------------------
// pass 1 (forward): per-chunk aggregate (P = prod a, S = state from zero) -> agg[b][j][c]
template <typename TIN, int VB, int LT, int NS>
__global__ void __launch_bounds__(TC * NS)
scan_fwd_state(const float *__restrict__ dlt, const float *__restrict__ A_log,
               const TIN *__restrict__ Bt, int64_t bt_rs, float2 *__restrict__ agg, ScanDims d) {
  constexpr int ROWB = TC * sizeof(TIN);
  constexpr int TS = LT / NS;
  constexpr int NTH = TC * NS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  TIN *bt = reinterpret_cast<TIN *>(smem);
  float *dl = reinterpret_cast<float *>(smem + LT * ROWB);
  float2 *segs = reinterpret_cast<float2 *>(smem + LT * ROWB + LT * d.HT * 4);

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
  stage_delta<LT, NTH>(dl, dlt, tok0, rows_valid, c0 >> d.log2N, (int)d.h, d.HT, d.softplus, tid);
  const float A2 = c < d.Dn ? -expf(A_log[c]) * LOG2E_F : 0.f;
  __syncthreads();

  float P = 1.f, S = 0.f;
  const int hh = lane >> d.log2N;
#pragma unroll
  for (int i = 0; i < TS; ++i) {
    int t = seg * TS + i;
    float a = __builtin_amdgcn_exp2f(dl[t * d.HT + hh] * A2);
    S = fmaf(a, S, to_f32(bt[t * TC + lane]));
    P *= a;
  }
  segs[seg * TC + lane] = make_float2(P, S);
  __syncthreads();
  if (seg == 0 && c < d.Dn) {
#pragma unroll
    for (int s = 1; s < NS; ++s) {
      float2 q = segs[s * TC + lane];
      S = fmaf(q.x, S, q.y);
      P *= q.x;
    }
    agg[((int64_t)b * d.nchunks + chunk) * d.Dn + c] = make_float2(P, S);
  }
}


}  // namespace
