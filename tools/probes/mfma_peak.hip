// Probe: sustained v_mfma_f32_16x16x32_bf16 rate with registers only (no LDS, no memory): the practical MFMA ceiling
// (clock under load) that the GEMM kernels' utilisation should be read against.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
template <int NACC>
__global__ void __launch_bounds__(512) k(float *out, int iters, int waves_active) {
  if ((int)(threadIdx.x >> 6) >= waves_active) return;
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  if (s == 123.456f) out[0] = s;
}
int main() {
  float *out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wa : {4, 8}) {
    for (int rep = 0; rep < 3; ++rep) {
      const int iters = 20000;
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<32>, dim3(256), dim3(512), 0, 0, out, iters, wa);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double fl = 256.0 * wa * iters * 32 * 16 * 16 * 32 * 2;
      printf("waves/CU=%d: %.2f ms  %.0f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", wa, ms, fl / ms / 1e9,
             ms * 1e-3 * 2.4e9 / (iters * 32.0 * wa / 4));
    }
  }
  return 0;
}
