// Next-token cross entropy on the logits as the LM head leaves them (reference core.py:1407-1416:
// shift_logits = logits[..., :-1, :], shift_labels = labels[..., 1:], CrossEntropyLoss(ignore_index=-100)
// in fp32).  The stock path materialises a contiguous copy of the shifted logits, an fp32 copy of that,
// the fp32 log-softmax, its fp32 gradient and a padded bf16 gradient - 35 ms of HBM traffic per step at
// [24, 4096, 32000].  Here the forward reads each logits row once (online max / sum-exp) and keeps only
// the row's log-sum-exp; the backward reads the row again and writes the gradient in the logits' dtype:
// 2 + 4 bytes per bf16 logit in total.  One 256-thread work-group per (batch, position) row.
#include "common.h"

namespace {

constexpr int CE_NT = 256;

template <typename T> struct ce_vec;
template <> struct ce_vec<bf16_t> { static constexpr int N = 8; };
template <> struct ce_vec<float> { static constexpr int N = 4; };

template <typename T> __device__ __forceinline__ void ce_unpack(const uint4 &v, float (&x)[ce_vec<T>::N]);
template <> __device__ __forceinline__ void ce_unpack<bf16_t>(const uint4 &v, float (&x)[8]) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    x[2 * i] = __builtin_bit_cast(float, w[i] << 16);
    x[2 * i + 1] = __builtin_bit_cast(float, w[i] & 0xffff0000u);
  }
}
template <> __device__ __forceinline__ void ce_unpack<float>(const uint4 &v, float (&x)[4]) {
  x[0] = __builtin_bit_cast(float, v.x); x[1] = __builtin_bit_cast(float, v.y);
  x[2] = __builtin_bit_cast(float, v.z); x[3] = __builtin_bit_cast(float, v.w);
}

template <typename T> __device__ __forceinline__ uint4 ce_pack(const float (&x)[ce_vec<T>::N]);
template <> __device__ __forceinline__ uint4 ce_pack<bf16_t>(const float (&x)[8]) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    w[i] = (uint32_t)__builtin_bit_cast(uint16_t, (bf16_t)x[2 * i]) | ((uint32_t)__builtin_bit_cast(uint16_t, (bf16_t)x[2 * i + 1]) << 16);
  return make_uint4(w[0], w[1], w[2], w[3]);
}
template <> __device__ __forceinline__ uint4 ce_pack<float>(const float (&x)[4]) {
  return make_uint4(__builtin_bit_cast(uint32_t, x[0]), __builtin_bit_cast(uint32_t, x[1]), __builtin_bit_cast(uint32_t, x[2]),
                    __builtin_bit_cast(uint32_t, x[3]));
}

// (m, s) = (running max, sum of exp(x - m)); the merge is associative and exact up to rounding
__device__ __forceinline__ void ce_merge(float &m, float &s, float m2, float s2) {
  const float mn = fmaxf(m, m2);
  if (mn == -INFINITY) { m = mn; s = 0.f; return; }
  s = s * __builtin_amdgcn_exp2f((m - mn) * LOG2E_F) + s2 * __builtin_amdgcn_exp2f((m2 - mn) * LOG2E_F);
  m = mn;
}

// target of logits row (b, l) is labels[b, l + 1]; rows l >= n_pos carry no loss
template <typename T>
__global__ void __launch_bounds__(CE_NT)
ce_fwd_k(const T *__restrict__ logits, const int64_t *__restrict__ labels, float *__restrict__ lse,
         float *__restrict__ row_loss, int L, int V, int label_stride, int n_pos, int64_t ignore_index) {
  constexpr int VN = ce_vec<T>::N;
  const int64_t row = blockIdx.x;
  const int b = (int)(row / L), l = (int)(row - (int64_t)b * L);
  int64_t target = ignore_index;
  if (l < n_pos) target = labels[(int64_t)b * label_stride + l + 1];
  if (target == ignore_index || target < 0 || target >= V) {   // uniform per work-group
    if (threadIdx.x == 0) { lse[row] = 0.f; row_loss[row] = 0.f; }
    return;
  }
  const T *x = logits + row * (int64_t)V;
  float m = -INFINITY, s = 0.f;
  for (int c = threadIdx.x; c < V / VN; c += CE_NT) {
    float v[VN];
    ce_unpack<T>(*reinterpret_cast<const uint4 *>(x + (int64_t)c * VN), v);
    float cm = v[0];
#pragma unroll
    for (int i = 1; i < VN; ++i) cm = fmaxf(cm, v[i]);
    float cs = 0.f;
#pragma unroll
    for (int i = 0; i < VN; ++i) cs += __builtin_amdgcn_exp2f((v[i] - cm) * LOG2E_F);
    ce_merge(m, s, cm, cs);
  }
  // wave, then work-group, in a fixed order
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float m2 = __shfl_xor(m, off), s2 = __shfl_xor(s, off);
    ce_merge(m, s, m2, s2);
  }
  __shared__ float sm[CE_NT / 64], ss[CE_NT / 64];
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = m; ss[threadIdx.x >> 6] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float M = sm[0], S = ss[0];
    for (int w = 1; w < CE_NT / 64; ++w) ce_merge(M, S, sm[w], ss[w]);
    const float r = M + logf(S);
    lse[row] = r;
    row_loss[row] = r - to_f32(x[target]);
  }
}

// dlogits[row] = (softmax(row) - onehot(target)) * gscale for rows with a target, 0 for the others
template <typename T>
__global__ void __launch_bounds__(CE_NT)
ce_bwd_k(const T *__restrict__ logits, const int64_t *__restrict__ labels, const float *__restrict__ lse,
         const float *__restrict__ gscale, T *__restrict__ dlogits, int L, int V, int label_stride, int n_pos,
         int64_t ignore_index) {
  constexpr int VN = ce_vec<T>::N;
  const int64_t row = blockIdx.x;
  const int b = (int)(row / L), l = (int)(row - (int64_t)b * L);
  int64_t target = ignore_index;
  if (l < n_pos) target = labels[(int64_t)b * label_stride + l + 1];
  const bool live = !(target == ignore_index || target < 0 || target >= V);
  const T *x = logits + row * (int64_t)V;
  T *dx = dlogits + row * (int64_t)V;
  if (!live) {
    for (int c = threadIdx.x; c < V / VN; c += CE_NT) *reinterpret_cast<uint4 *>(dx + (int64_t)c * VN) = make_uint4(0, 0, 0, 0);
    return;
  }
  const float g = gscale[0], r = lse[row];
  const int tc = (int)(target / VN), ti = (int)(target - (int64_t)tc * VN);
  for (int c = threadIdx.x; c < V / VN; c += CE_NT) {
    float v[VN];
    ce_unpack<T>(*reinterpret_cast<const uint4 *>(x + (int64_t)c * VN), v);
#pragma unroll
    for (int i = 0; i < VN; ++i) {
      float p = __builtin_amdgcn_exp2f((v[i] - r) * LOG2E_F);
      if (c == tc && i == ti) p -= 1.f;
      v[i] = p * g;
    }
    *reinterpret_cast<uint4 *>(dx + (int64_t)c * VN) = ce_pack<T>(v);
  }
}

// Forward AND backward of a row in one pass (round 6; the fused LM head + loss, which knows the gradient's scale before the
// forward: ops/loss.py).  The row stays in registers between the two sweeps - V / VN chunks of 16 bytes over 256 threads,
// at most CE_KEEP per thread (V <= 32768 bf16 / 16384 fp32 logits) - so the logits are read from HBM once instead of twice:
// 2 + 2 bytes per bf16 logit instead of 2 + 4.  The first sweep is ce_fwd_k's (same chunk order per thread, same merges in
// the same order: the same log-sum-exp bit for bit), the second ce_bwd_k's on the kept values.  dlogits may alias logits.
constexpr int CE_KEEP = 16;

template <typename T>
__global__ void __launch_bounds__(CE_NT)
ce_fwd_bwd_k(const T *logits, const int64_t *__restrict__ labels, float *__restrict__ lse,
             float *__restrict__ row_loss, const float *__restrict__ gscale, T *dlogits, int L, int V,
             int label_stride, int n_pos, int64_t ignore_index) {
  constexpr int VN = ce_vec<T>::N;
  const int64_t row = blockIdx.x;
  const int b = (int)(row / L), l = (int)(row - (int64_t)b * L);
  int64_t target = ignore_index;
  if (l < n_pos) target = labels[(int64_t)b * label_stride + l + 1];
  const T *x = logits + row * (int64_t)V;
  T *dx = dlogits + row * (int64_t)V;
  const int nch = V / VN;
  if (target == ignore_index || target < 0 || target >= V) {   // uniform per work-group
    if (threadIdx.x == 0) { lse[row] = 0.f; row_loss[row] = 0.f; }
    for (int c = threadIdx.x; c < nch; c += CE_NT) *reinterpret_cast<uint4 *>(dx + (int64_t)c * VN) = make_uint4(0, 0, 0, 0);
    return;
  }
  uint4 keep[CE_KEEP];
  float m = -INFINITY, s = 0.f;
#pragma unroll
  for (int k = 0; k < CE_KEEP; ++k) {
    const int c = (int)threadIdx.x + k * CE_NT;
    if (c < nch) {
      keep[k] = *reinterpret_cast<const uint4 *>(x + (int64_t)c * VN);
      float v[VN];
      ce_unpack<T>(keep[k], v);
      float cm = v[0];
#pragma unroll
      for (int i = 1; i < VN; ++i) cm = fmaxf(cm, v[i]);
      float cs = 0.f;
#pragma unroll
      for (int i = 0; i < VN; ++i) cs += __builtin_amdgcn_exp2f((v[i] - cm) * LOG2E_F);
      ce_merge(m, s, cm, cs);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float m2 = __shfl_xor(m, off), s2 = __shfl_xor(s, off);
    ce_merge(m, s, m2, s2);
  }
  __shared__ float sm[CE_NT / 64], ss[CE_NT / 64], sr;
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = m; ss[threadIdx.x >> 6] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float M = sm[0], S = ss[0];
    for (int w = 1; w < CE_NT / 64; ++w) ce_merge(M, S, sm[w], ss[w]);
    const float r0 = M + logf(S);
    lse[row] = r0;
    row_loss[row] = r0 - to_f32(x[target]);   // (read before any thread overwrites the row: the barrier below)
    sr = r0;
  }
  __syncthreads();
  const float g = gscale[0], r = sr;
  const int tc = (int)(target / VN), ti = (int)(target - (int64_t)tc * VN);
#pragma unroll
  for (int k = 0; k < CE_KEEP; ++k) {
    const int c = (int)threadIdx.x + k * CE_NT;
    if (c < nch) {
      float v[VN];
      ce_unpack<T>(keep[k], v);
#pragma unroll
      for (int i = 0; i < VN; ++i) {
        float p = __builtin_amdgcn_exp2f((v[i] - r) * LOG2E_F);
        if (c == tc && i == ti) p -= 1.f;
        v[i] = p * g;
      }
      *reinterpret_cast<uint4 *>(dx + (int64_t)c * VN) = ce_pack<T>(v);
    }
  }
}

}  // namespace

extern "C" int apertis_cross_entropy_fwd(const void *logits, const int64_t *labels, float *lse, float *row_loss,
                                         int64_t B, int64_t L, int64_t V, int64_t label_stride, int64_t n_pos,
                                         int64_t ignore_index, int dtype, void *stream) {
  if (!logits || !labels || !lse || !row_loss || B < 0 || L <= 0 || V <= 0 || n_pos < 0) return APERTIS_ERR_ARG;
  if (n_pos > L || n_pos + 1 > label_stride || B * L > 0x7fffffffLL || V > 0x7fffffffLL) return APERTIS_ERR_ARG;
  if (B == 0) return APERTIS_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == APERTIS_BF16) {
    if (V % 8 || (((uintptr_t)logits) & 15)) return APERTIS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(ce_fwd_k<bf16_t>, dim3((unsigned)(B * L)), dim3(CE_NT), 0, st, (const bf16_t *)logits, labels, lse,
                       row_loss, (int)L, (int)V, (int)label_stride, (int)n_pos, ignore_index);
  } else if (dtype == APERTIS_F32) {
    if (V % 4 || (((uintptr_t)logits) & 15)) return APERTIS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(ce_fwd_k<float>, dim3((unsigned)(B * L)), dim3(CE_NT), 0, st, (const float *)logits, labels, lse,
                       row_loss, (int)L, (int)V, (int)label_stride, (int)n_pos, ignore_index);
  } else {
    return APERTIS_ERR_ARG;
  }
  return apertis_check_launch();
}

extern "C" int apertis_cross_entropy_bwd(const void *logits, const int64_t *labels, const float *lse, const float *gscale,
                                         void *dlogits, int64_t B, int64_t L, int64_t V, int64_t label_stride,
                                         int64_t n_pos, int64_t ignore_index, int dtype, void *stream) {
  if (!logits || !labels || !lse || !gscale || !dlogits || B < 0 || L <= 0 || V <= 0 || n_pos < 0) return APERTIS_ERR_ARG;
  if (n_pos > L || n_pos + 1 > label_stride || B * L > 0x7fffffffLL || V > 0x7fffffffLL) return APERTIS_ERR_ARG;
  if (B == 0) return APERTIS_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == APERTIS_BF16) {
    if (V % 8 || (((uintptr_t)logits) & 15) || (((uintptr_t)dlogits) & 15)) return APERTIS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(ce_bwd_k<bf16_t>, dim3((unsigned)(B * L)), dim3(CE_NT), 0, st, (const bf16_t *)logits, labels, lse,
                       gscale, (bf16_t *)dlogits, (int)L, (int)V, (int)label_stride, (int)n_pos, ignore_index);
  } else if (dtype == APERTIS_F32) {
    if (V % 4 || (((uintptr_t)logits) & 15) || (((uintptr_t)dlogits) & 15)) return APERTIS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(ce_bwd_k<float>, dim3((unsigned)(B * L)), dim3(CE_NT), 0, st, (const float *)logits, labels, lse,
                       gscale, (float *)dlogits, (int)L, (int)V, (int)label_stride, (int)n_pos, ignore_index);
  } else {
    return APERTIS_ERR_ARG;
  }
  return apertis_check_launch();
}

// Both passes in one launch for callers that know the gradient's scale up front: lse / row_loss as apertis_cross_entropy_fwd
// leaves them and dlogits (which may be `logits` itself) as apertis_cross_entropy_bwd does, bit for bit, with one read of the
// logits.  APERTIS_ERR_UNSUPPORTED when a row does not fit the registers of its work-group (V > 32768 bf16 / 16384 fp32) or
// the two-pass forms' alignment rules fail: call the two entry points instead.
extern "C" int apertis_cross_entropy_fwd_bwd(const void *logits, const int64_t *labels, float *lse, float *row_loss,
                                             const float *gscale, void *dlogits, int64_t B, int64_t L, int64_t V,
                                             int64_t label_stride, int64_t n_pos, int64_t ignore_index, int dtype, void *stream) {
  if (!logits || !labels || !lse || !row_loss || !gscale || !dlogits || B < 0 || L <= 0 || V <= 0 || n_pos < 0) return APERTIS_ERR_ARG;
  if (n_pos > L || n_pos + 1 > label_stride || B * L > 0x7fffffffLL || V > 0x7fffffffLL) return APERTIS_ERR_ARG;
  if (B == 0) return APERTIS_OK;
  hipStream_t st = (hipStream_t)stream;
  if ((((uintptr_t)logits) & 15) || (((uintptr_t)dlogits) & 15)) return APERTIS_ERR_UNSUPPORTED;
  if (dtype == APERTIS_BF16) {
    if (V % 8 || V / 8 > CE_KEEP * CE_NT) return APERTIS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(ce_fwd_bwd_k<bf16_t>, dim3((unsigned)(B * L)), dim3(CE_NT), 0, st, (const bf16_t *)logits, labels, lse,
                       row_loss, gscale, (bf16_t *)dlogits, (int)L, (int)V, (int)label_stride, (int)n_pos, ignore_index);
  } else if (dtype == APERTIS_F32) {
    if (V % 4 || V / 4 > CE_KEEP * CE_NT) return APERTIS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(ce_fwd_bwd_k<float>, dim3((unsigned)(B * L)), dim3(CE_NT), 0, st, (const float *)logits, labels, lse,
                       row_loss, gscale, (float *)dlogits, (int)L, (int)V, (int)label_stride, (int)n_pos, ignore_index);
  } else {
    return APERTIS_ERR_ARG;
  }
  return apertis_check_launch();
}
