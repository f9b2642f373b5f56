// Optimizer step of the training path: global gradient-norm clipping + AdamW in two passes over the gradients
// instead of four passes over gradients and one over everything (reference src/training/pipeline.py:469-473
// AdamW parameter groups, :544-546 clip_grad_norm_ -> optimizer.step; torch.optim.AdamW's update rule).
//   pass 1  apertis_grad_sumsq      per-chunk sums of g^2 (fixed order inside a chunk)
//           apertis_clip_coef       folds the chunk sums in index order -> ||g||, coef = min(1, max_norm/(||g||+1e-6))
//   pass 2  apertis_adamw_step      p, m, v updated in place from g*coef (the gradients are not written back)
// HBM-bound: 4 B read per parameter in pass 1, 16 B read + 12 B written in pass 2.  A "chunk" is 16384 consecutive
// elements of one tensor (one work-group); tensors are described by a device table of pointers so one launch covers a
// whole parameter group.  fp32 parameters, gradients and moments.
#include "common.h"

namespace {

constexpr int OPT_NT = 256;
constexpr int OPT_CHUNK = 16384;   // elements per work-group = 16 float4 per thread

struct OptTensor {   // mirrors apertis_opt_tensor in include/apertis_hip.h
  float *p, *g, *m, *v;
  int64_t numel;
};

__device__ __forceinline__ float block_sum_256(float v, float *sm) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__global__ void __launch_bounds__(OPT_NT)
grad_sumsq_k(const OptTensor *__restrict__ tensors, const int32_t *__restrict__ chunk_tensor,
             const int32_t *__restrict__ chunk_index, float *__restrict__ partials) {
  __shared__ float sm[4];
  const OptTensor t = tensors[chunk_tensor[blockIdx.x]];
  const int64_t base = (int64_t)chunk_index[blockIdx.x] * OPT_CHUNK;
  const int64_t n = min((int64_t)OPT_CHUNK, t.numel - base);
  const float *g = t.g + base;
  float acc = 0.f;
  if ((((uintptr_t)g) & 15) == 0) {
    const int64_t n4 = n >> 2;
    for (int64_t i = threadIdx.x; i < n4; i += OPT_NT) {
      const float4 x = reinterpret_cast<const float4 *>(g)[i];
      acc += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
    }
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += OPT_NT) acc += g[i] * g[i];
  } else {
    for (int64_t i = threadIdx.x; i < n; i += OPT_NT) acc += g[i] * g[i];
  }
  const float s = block_sum_256(acc, sm);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// one work-group: sum of the partials in index order (double accumulation), then norm and clip coefficient
__global__ void __launch_bounds__(1024) clip_coef_k(const float *__restrict__ partials, int64_t n, float max_norm,
                                                    float *__restrict__ out, const int32_t *__restrict__ poison) {
  __shared__ double sm[1024];
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) acc += (double)partials[i];
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(sm[0]);
    out[0] = norm;
    // clip_grad_norm_: coef = clamp(max / (norm + 1e-6), max = 1); torch.clamp hands a NaN through (fminf would
    // turn it into 1 and apply an unclipped step on non-finite gradients)
    const float c = max_norm / (norm + 1e-6f);
    out[1] = (c != c) ? c : fminf(c, 1.f);
    // a non-zero poison word (the single-pass scan's look-back time-out flag, scan_gate.hip: the step's activations are
    // wrong) rejects the step visibly - the norm reads NaN - and WITHOUT destroying the model: the coefficient -1 makes the
    // AdamW pass a no-op (a transient time-out must not overwrite every parameter and moment with NaN)
    if (poison && *poison != 0) { out[0] = __builtin_nanf(""); out[1] = -1.f; }
  }
}

// every derived constant is formed in double on the host and rounded once, as the Python reference does
struct AdamArgs { float decay, one_minus_beta1, beta2, one_minus_beta2, eps, step_size, bc2_sqrt; };

__device__ __forceinline__ void adamw_one(float &p, float g, float &m, float &v, const AdamArgs &a, float coef) {
  g *= coef;
  p *= a.decay;                                     // decoupled weight decay: 1 - lr * wd
  m += (g - m) * a.one_minus_beta1;                 // exp_avg.lerp_(grad, 1 - beta1)
  v = v * a.beta2 + a.one_minus_beta2 * g * g;      // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  p -= a.step_size * (m / denom);                   // step_size = lr / (1 - beta1^step)
}

typedef __attribute__((ext_vector_type(4))) float opt_f4;
__device__ __forceinline__ float4 ntload4(const float *p) {
  const opt_f4 t = __builtin_nontemporal_load(reinterpret_cast<const opt_f4 *>(p));
  return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void ntstore4(float *p, float4 v) {
  opt_f4 o = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(o, reinterpret_cast<opt_f4 *>(p));
}

__global__ void __launch_bounds__(OPT_NT)
adamw_step_k(const OptTensor *__restrict__ tensors, const int32_t *__restrict__ chunk_tensor,
             const int32_t *__restrict__ chunk_index, AdamArgs a, const float *__restrict__ coef_ptr) {
  const OptTensor t = tensors[chunk_tensor[blockIdx.x]];
  const int64_t base = (int64_t)chunk_index[blockIdx.x] * OPT_CHUNK;
  const int64_t n = min((int64_t)OPT_CHUNK, t.numel - base);
  const float coef = coef_ptr ? coef_ptr[1] : 1.f;
  if (coef < 0.f) return;   // (clip_coef's "skip this step": the poison word was set)
  float *p = t.p + base, *m = t.m + base, *v = t.v + base;
  const float *g = t.g + base;
  const bool vec = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
  int64_t done = 0;
  if (vec) {
    const int64_t n4 = n >> 2;
    for (int64_t i = threadIdx.x; i < n4; i += OPT_NT) {
      // every element is read once and written once per step: non-temporal both ways (nothing here is worth a cache line)
      float4 pp = ntload4(p + 4 * i), mm = ntload4(m + 4 * i), vv = ntload4(v + 4 * i);
      const float4 gg = ntload4(g + 4 * i);
      adamw_one(pp.x, gg.x, mm.x, vv.x, a, coef); adamw_one(pp.y, gg.y, mm.y, vv.y, a, coef);
      adamw_one(pp.z, gg.z, mm.z, vv.z, a, coef); adamw_one(pp.w, gg.w, mm.w, vv.w, a, coef);
      ntstore4(p + 4 * i, pp); ntstore4(m + 4 * i, mm); ntstore4(v + 4 * i, vv);
    }
    done = n4 << 2;
  }
  for (int64_t i = done + threadIdx.x; i < n; i += OPT_NT) adamw_one(p[i], g[i], m[i], v[i], a, coef);
}

}  // namespace

extern "C" int64_t apertis_opt_chunk_elems(void) { return OPT_CHUNK; }

extern "C" int apertis_grad_sumsq(const void *tensors, const int32_t *chunk_tensor, const int32_t *chunk_index,
                                  int64_t n_chunks, float *partials, void *stream) {
  if (n_chunks < 0 || n_chunks > 0x7fffffffLL) return APERTIS_ERR_ARG;
  if (n_chunks == 0) return APERTIS_OK;
  if (!tensors || !chunk_tensor || !chunk_index || !partials) return APERTIS_ERR_ARG;
  hipLaunchKernelGGL(grad_sumsq_k, dim3((unsigned)n_chunks), dim3(OPT_NT), 0, (hipStream_t)stream,
                     (const OptTensor *)tensors, chunk_tensor, chunk_index, partials);
  return apertis_check_launch();
}

extern "C" int apertis_clip_coef(const float *partials, int64_t n, float max_norm, float *norm_coef, const int32_t *poison,
                                 void *stream) {
  if (!norm_coef || n < 0 || (n > 0 && !partials) || !(max_norm >= 0.f)) return APERTIS_ERR_ARG;
  hipLaunchKernelGGL(clip_coef_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, partials, n, max_norm, norm_coef, poison);
  return apertis_check_launch();
}

extern "C" int apertis_adamw_step(const void *tensors, const int32_t *chunk_tensor, const int32_t *chunk_index,
                                  int64_t n_chunks, double lr, double beta1, double beta2, double eps, double weight_decay,
                                  int64_t step, const float *norm_coef, void *stream) {
  if (n_chunks < 0 || n_chunks > 0x7fffffffLL || step < 1) return APERTIS_ERR_ARG;
  if (n_chunks == 0) return APERTIS_OK;
  if (!tensors || !chunk_tensor || !chunk_index) return APERTIS_ERR_ARG;
  AdamArgs a;
  a.decay = (float)(1.0 - (double)lr * (double)weight_decay);
  a.one_minus_beta1 = (float)(1.0 - beta1);
  a.beta2 = (float)beta2;
  a.one_minus_beta2 = (float)(1.0 - beta2);
  a.eps = (float)eps;
  a.step_size = (float)((double)lr / (1.0 - pow(beta1, (double)step)));
  a.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
  hipLaunchKernelGGL(adamw_step_k, dim3((unsigned)n_chunks), dim3(OPT_NT), 0, (hipStream_t)stream,
                     (const OptTensor *)tensors, chunk_tensor, chunk_index, a, norm_coef);
  return apertis_check_launch();
}
