// Single-token decode step of the SSM block, re-ordered (round 5).
//
// Reference: SelectiveLinearAttention.forward with a cache and L = 1, /root/reference/src/model/core.py:364-400 (called by
// generate(), core.py:1578-1603).  The reference prepends the cached conv window to the new token's xp and keeps the FIRST
// L = 1 conv outputs (core.py:369-373) - an output that sees k - 1 zeros and window[0] only (the front-slice quirk this repo
// reproduces: SURVEY 3.3, scan_gate.hip decode_conv_k).  So the conv output of a token step, and with it x_param_proj, the dt
// projection, the state update and  C s + D xc,  depend on the CACHES alone - not on the layer's input at this step; only the
// gate silu(z) and the value pushed into the window come from the current token.  That part of every layer can therefore run
// at the START of the token step, all layers in one launch each (the layers side by side as one more batch dimension):
//   decode_pre_conv_k   xc[l, b, c] = silu(w[l, c, k-1] window[l, b, c, 0] + bias[l, c])        (decode_conv_k's arithmetic)
//   (x_param_proj: the grouped skinny NT kernel with one group per layer - grouped_gemm.hip)
//   decode_pre_state_k  dt_proj_head + softplus, s <- exp(delta A) s + Bt (in place),  pre = C s + D xc   (decode_state_k's)
// and inside the layer loop only
//   decode_post_k       gated = pre * silu(z)  and the window push  [w1 .. w_{k-2}, xp]
// is left between in_proj and out_proj: three dependent launches per layer and token (conv, x_param GEMV, state) become three
// launches per token.  Same arithmetic, operation for operation, as the per-layer kernels: bit-identical outputs and caches.
#include "scan_lean.h"

namespace {

template <typename T>
__global__ void __launch_bounds__(256)
decode_pre_conv_k(const T *__restrict__ conv_state, const float *__restrict__ w, const float *__restrict__ bias, T *__restrict__ xc,
                  int64_t NL, int64_t B, int64_t Dn, int k) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= NL * B * Dn) return;
  const int64_t l = i / (B * Dn), c = i % Dn;
  const float first = to_f32(conv_state[i * (k - 1)]);
  const float acc = w[(l * Dn + c) * k + (k - 1)] * first + bias[l * Dn + c];
  xc[i] = from_f32<T>(acc / (1.f + expf(-acc)));
}

template <typename T>
__global__ void __launch_bounds__(256)
decode_pre_state_k(const T *__restrict__ p, int64_t p_rs, int64_t off_bt, int64_t off_c, int64_t off_dt, const float *__restrict__ Wdt,
                   const float *__restrict__ bdt, int R, const float *__restrict__ A_log, const float *__restrict__ Dv,
                   const T *__restrict__ xc, float *__restrict__ state, float *__restrict__ pre, int64_t NL, int64_t B, int64_t h,
                   int64_t N, int softplus) {
  const int64_t Dn = h * N;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= NL * B * Dn) return;
  const int64_t l = i / (B * Dn), c = i % Dn, row = i / Dn, hd = c / N;
  const T *pr = p + row * p_rs;
  const T *xr = pr + off_dt;
  const float *w = Wdt + (l * h + hd) * R;
  float dlv = bdt ? bdt[l * h + hd] : 0.f;
  for (int r = 0; r < R; ++r) dlv = fmaf(to_f32(xr[r]), w[r], dlv);
  if (softplus) dlv = softplus_f(dlv);
  const float av = __builtin_amdgcn_exp2f(dlv * (-expf(A_log[l * Dn + c]) * LOG2E_F));
  const float s = fmaf(av, state[i], to_f32(pr[off_bt + c]));
  state[i] = s;
  const float yv = to_f32(pr[off_c + c]) * s;
  const float dx = Dv[l * Dn + c] * to_f32(xc[i]);
  pre[i] = yv + dx;
}

template <typename T>
__global__ void __launch_bounds__(256)
decode_post_k(const float *__restrict__ pre, const T *__restrict__ xz, int64_t xz_rs, T *conv_state, T *__restrict__ gated, int64_t B,
              int64_t Dn, int k) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B * Dn) return;
  const int64_t b = i / Dn, c = i - b * Dn;
  gated[i] = from_f32<T>(pre[i] * silu_g(to_f32(xz[b * xz_rs + Dn + c])));
  // the window push (decode_conv_k's second half): the last k - 1 tokens of [window | xp]
  T *cs = conv_state + i * (k - 1);
  constexpr int KEEP = 14;
  T keep[KEEP];
#pragma unroll
  for (int j = 0; j < KEEP; ++j) keep[j] = j + 1 < k - 1 ? cs[j + 1] : T(0);
  const T last = xz[b * xz_rs + c];
#pragma unroll
  for (int j = 0; j < KEEP; ++j)
    if (j + 1 < k - 1) cs[j] = keep[j];
  cs[k - 2] = last;
}

// out [B, N] = x W^T (+ bias) for a handful of rows with the row count by VALUE: grouped_gemm_nt_skinny_k's K < 512 form
// (grouped_gemm.hip: a wave owns 16 output columns, W rows on the MFMA A operand straight from global memory, the rows of x on
// B, eight 32-deep steps per batch in that order, one 8-byte store per lane - the same bits) without the load of the group
// offsets in front of everything else: one dependent round trip less in a kernel that is a chain of three.
typedef __attribute__((ext_vector_type(8))) bf16_t ds_bf16x8;
typedef __attribute__((ext_vector_type(4))) float ds_f32x4;

__global__ void __launch_bounds__(64)
decode_dense_gemv_k(const bf16_t *__restrict__ x, const bf16_t *__restrict__ W, int ldw, const float *__restrict__ bias,
                    bf16_t *__restrict__ out, int B, int K, int N) {
  constexpr int U = 8;
  const int lane = threadIdx.x, n0 = blockIdx.x * 16, l15 = lane & 15, fg = lane >> 4, kc = fg * 8;
  const int wcol = n0 + l15;
  const bf16_t *wrow = W + (int64_t)min(wcol, N - 1) * ldw;
  const bool w_ok = wcol < N;
  const ds_bf16x8 zero = {};
  const int nq = n0 + fg * 4;
  float bq[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[q] = nq + q < N ? bias[nq + q] : 0.f;
  }
  const bool x_ok = l15 < B;
  const bf16_t *xrow = x + (int64_t)min(l15, B - 1) * K;
  ds_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 32 * U) {
    ds_bf16x8 a[U], bb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kk = k0 + u * 32 + kc;
      const bool ok = kk < K;
      a[u] = (ok && w_ok) ? *reinterpret_cast<const ds_bf16x8 *>(wrow + kk) : zero;
      bb[u] = (ok && x_ok) ? *reinterpret_cast<const ds_bf16x8 *>(xrow + kk) : zero;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], bb[u], acc, 0, 0, 0);
  }
  if (x_ok && nq < N) {
    bf16_t o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = from_f32<bf16_t>(acc[q] + bq[q]);
    bf16_t *dst = out + (int64_t)l15 * N + nq;
    if (nq + 3 < N) *reinterpret_cast<uint2 *>(dst) = *reinterpret_cast<const uint2 *>(o);
    else for (int q = 0; q < 4 && nq + q < N; ++q) dst[q] = o[q];
  }
}

}  // namespace

extern "C" int apertis_decode_dense_gemv(const void *x, const void *W, int64_t ldw, const float *bias, void *out, int64_t B, int64_t K,
                                         int64_t N, void *stream) {
  if (!x || !W || !out || ldw < K) return APERTIS_ERR_ARG;
  if (B < 1 || B > 16 || K < 8 || K % 8 || K >= 512 || N < 4 || N % 4 || ldw % 8) return APERTIS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(decode_dense_gemv_k, dim3((unsigned)ceil_div64(N, 16)), dim3(64), 0, (hipStream_t)stream, (const bf16_t *)x,
                     (const bf16_t *)W, (int)ldw, bias, (bf16_t *)out, (int)B, (int)K, (int)N);
  return apertis_check_launch();
}

extern "C" int apertis_decode_pre_conv(const void *conv_state, const float *w, const float *bias, void *xc, int64_t NL, int64_t B,
                                       int64_t Dn, int64_t k, int dtype, void *stream) {
  if (!conv_state || !w || !bias || !xc || NL <= 0 || B <= 0 || Dn <= 0) return APERTIS_ERR_ARG;
  if (k < 2 || k > 16) return APERTIS_ERR_UNSUPPORTED;       // (k = 1: the conv output is the new token's own xp)
  const unsigned grid = (unsigned)ceil_div64(NL * B * Dn, 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == APERTIS_F32)
    hipLaunchKernelGGL(decode_pre_conv_k<float>, dim3(grid), dim3(256), 0, st, (const float *)conv_state, w, bias, (float *)xc, NL, B, Dn, (int)k);
  else if (dtype == APERTIS_BF16)
    hipLaunchKernelGGL(decode_pre_conv_k<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t *)conv_state, w, bias, (bf16_t *)xc, NL, B, Dn, (int)k);
  else
    return APERTIS_ERR_UNSUPPORTED;
  return apertis_check_launch();
}

extern "C" int apertis_decode_pre_state(const void *p, int64_t p_rs, int64_t off_bt, int64_t off_c, int64_t off_dt, const float *W_dt,
                                        const float *b_dt, int64_t R, const float *A_log, const float *D, const void *xc, float *state,
                                        float *pre, int64_t NL, int64_t B, int64_t h, int64_t N, int delta_softplus, int dtype,
                                        void *stream) {
  if (!p || !W_dt || !A_log || !D || !xc || !state || !pre || NL <= 0 || B <= 0 || h <= 0 || N <= 0 || R < 1) return APERTIS_ERR_ARG;
  const int64_t Dn = h * N;
  if (off_bt < 0 || off_c < 0 || off_dt < 0 || off_bt + Dn > p_rs || off_c + Dn > p_rs || off_dt + R > p_rs) return APERTIS_ERR_ARG;
  const unsigned grid = (unsigned)ceil_div64(NL * B * Dn, 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == APERTIS_F32)
    hipLaunchKernelGGL(decode_pre_state_k<float>, dim3(grid), dim3(256), 0, st, (const float *)p, p_rs, off_bt, off_c, off_dt, W_dt, b_dt,
                       (int)R, A_log, D, (const float *)xc, state, pre, NL, B, h, N, delta_softplus);
  else if (dtype == APERTIS_BF16)
    hipLaunchKernelGGL(decode_pre_state_k<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t *)p, p_rs, off_bt, off_c, off_dt, W_dt, b_dt,
                       (int)R, A_log, D, (const bf16_t *)xc, state, pre, NL, B, h, N, delta_softplus);
  else
    return APERTIS_ERR_UNSUPPORTED;
  return apertis_check_launch();
}

extern "C" int apertis_decode_post(const float *pre, const void *xz, int64_t xz_rs, void *conv_state, void *gated, int64_t B, int64_t Dn,
                                   int64_t k, int dtype, void *stream) {
  if (!pre || !xz || !conv_state || !gated || B <= 0 || Dn <= 0 || xz_rs < 2 * Dn) return APERTIS_ERR_ARG;
  if (k < 2 || k > 16) return APERTIS_ERR_UNSUPPORTED;
  const unsigned grid = (unsigned)ceil_div64(B * Dn, 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == APERTIS_F32)
    hipLaunchKernelGGL(decode_post_k<float>, dim3(grid), dim3(256), 0, st, pre, (const float *)xz, xz_rs, (float *)conv_state, (float *)gated, B, Dn, (int)k);
  else if (dtype == APERTIS_BF16)
    hipLaunchKernelGGL(decode_post_k<bf16_t>, dim3(grid), dim3(256), 0, st, pre, (const bf16_t *)xz, xz_rs, (bf16_t *)conv_state, (bf16_t *)gated, B, Dn, (int)k);
  else
    return APERTIS_ERR_UNSUPPORTED;
  return apertis_check_launch();
}
