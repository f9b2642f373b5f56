// Probe: does a VALU wave make progress beside an MFMA wave on the same SIMD?  One work-group of 8 waves per CU (two per
// SIMD): waves 0-3 run a dense MFMA loop (operands in registers, 8 independent accumulator tiles), waves 4-7 a VALU loop of
// the saved-gradient epilogue's mix (v_pk_fma_f32 / v_pk_mul_f32 / v_exp_f32 / v_rcp_f32 / integer).  Timed: MFMA waves
// alone, VALU waves alone, both.  MODE 0: v_mfma_f32_16x16x32_bf16, 1: v_mfma_f32_32x32x16_bf16 (same FLOPs per loop trip).
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_overlap.hip -o tools/probes/mfma_valu_overlap.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float v2f;

template <int MODE>
__global__ void __launch_bounds__(512) overlap_k(float *out, int iters, int run_mfma, int run_valu, unsigned seed) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float sink = 0.f;
  if (wave < 4) {
    if (!run_mfma) return;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) {
      unsigned h = (lane * 8 + i) * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u;
      a[i] = (__bf16)(((int)(h & 0xffff) - 32768) / 32768.f);
      b[i] = (__bf16)(((int)(h >> 16) - 32768) / 32768.f);
    }
    if constexpr (MODE == 0) {
      f32x4 acc[16];
      for (int j = 0; j < 16; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
      }
      for (int j = 0; j < 16; ++j) sink += acc[j][0];
    } else {
      f32x16 acc[4];
      for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
      }
      for (int j = 0; j < 4; ++j) sink += acc[j][0];
    }
  } else {
    if (!run_valu) return;
    v2f x[4], y[4];
    unsigned h[4];
    for (int w = 0; w < 4; ++w) { x[w] = (v2f){lane * 0.01f + w, lane * 0.02f - w}; y[w] = x[w]; h[w] = lane * 77u + w; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {          // ~ the epilogue's mix per element pair: 6 packed fp32, 2 exp, 2 rcp, 6 integer
          v2f t = __builtin_elementwise_fma(x[w], (v2f){0.33f, 0.33f}, (v2f){1.f, 1.f});
          t = (v2f){__builtin_amdgcn_rcpf(t.x), __builtin_amdgcn_rcpf(t.y)};
          v2f e = x[w] * x[w];
          e = (v2f){__builtin_amdgcn_exp2f(-e.x), __builtin_amdgcn_exp2f(-e.y)};
          v2f p = __builtin_elementwise_fma(t, (v2f){0.7f, 0.7f}, (v2f){-0.1f, -0.1f});
          p = __builtin_elementwise_fma(p, t, (v2f){0.35f, 0.35f});
          y[w] = __builtin_elementwise_fma(p * t, e, y[w] * (v2f){0.5f, 0.5f});
          h[w] ^= h[w] >> 16; h[w] *= 0x85ebca6bu; h[w] ^= h[w] >> 13; h[w] *= 0xc2b2ae35u; h[w] ^= h[w] >> 16;
          x[w].x += (h[w] & 1) * 1e-6f;
        }
      }
    }
    for (int w = 0; w < 4; ++w) sink += y[w].x + y[w].y + x[w].x;
  }
  if (sink == 12345.678f) out[0] = sink;
}
template <int MODE> float run(float *out, int iters, int m, int v) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  overlap_k<MODE><<<256, 512>>>(out, iters, m, v, 1u); hipDeviceSynchronize();
  float best = 1e9;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0); overlap_k<MODE><<<256, 512>>>(out, iters, m, v, 7u + r); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
  }
  return best * 1e3f;
}
int main() {
  float *out; hipMalloc(&out, 64);
  const int iters = 4000;
  // per loop trip and MFMA wave: 16 x 16x16x32 = 8 x 32x32x16 = 131072 MACs; VALU wave: 8 element pairs
  for (int mode = 0; mode < 2; ++mode) {
    float tm = mode ? run<1>(out, iters, 1, 0) : run<0>(out, iters, 1, 0);
    float tv = mode ? run<1>(out, iters, 0, 1) : run<0>(out, iters, 0, 1);
    float tb = mode ? run<1>(out, iters, 1, 1) : run<0>(out, iters, 1, 1);
    const double tf = 2.0 * 131072.0 * iters * 4 * 256 / (tm * 1e-6) / 1e12;
    printf("%s: MFMA waves alone %8.1f us (%6.0f TF chip-wide), VALU waves alone %8.1f us, both %8.1f us  (sum %8.1f, max %8.1f)\n",
           mode ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_16x16x32_bf16", tm, tf, tv, tb, tm + tv, tm > tv ? tm : tv);
  }
  return 0;
}
