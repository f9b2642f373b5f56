// Probe (round 6): does the SAME wave hide its own VALU work between its MFMAs?  One wave per SIMD (256 threads per work-group,
// one work-group per CU: the 512-register regime), a loop trip = 32 v_mfma_f32_16x16x32_bf16 on 32 independent accumulator
// tiles (a 64 x 128 wave tile's sub-step) with V VALU instructions of the saved-gradient epilogue's mix spread between the
// groups of four MFMAs by sched_group_barrier.  r4's probe (mfma_valu_overlap.hip) put the VALU work in ANOTHER wave of the
// SIMD and found the times to ADD; MI355X_MICROARCH.md's constants say an MFMA holds the SIMD's issue port for 8 of its 16
// cycles and fillers cost their issue slots (4 cycles, transcendentals 8).  Timed per V: MFMAs alone, VALU alone, interleaved.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_samewave.hip -o tools/probes/mfma_valu_samewave.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float v2f;

// one "unit" = 8 VALU instructions of the epilogue's mix (5 packed / plain fp32, 1 rcp, 1 exp2, 1 integer)
__device__ __forceinline__ void unit(v2f &x, v2f &y, unsigned &h) {
  v2f t = __builtin_elementwise_fma(x, (v2f){0.33f, 0.33f}, (v2f){1.f, 1.f});
  t.x = __builtin_amdgcn_rcpf(t.x);
  v2f e = x * x;
  e.x = __builtin_amdgcn_exp2f(-e.x);
  v2f p = __builtin_elementwise_fma(t, (v2f){0.7f, 0.7f}, (v2f){-0.1f, -0.1f});
  p = __builtin_elementwise_fma(p, t, (v2f){0.35f, 0.35f});
  y = __builtin_elementwise_fma(p, e, y);
  h = h * 2654435761u;
  x.y += __uint_as_float((h >> 9) | 0x3f800000u) * 1e-9f;
}

template <int UNITS, bool MF, bool VA>   // UNITS per group of four MFMAs
__global__ void __launch_bounds__(256) k(float *out, int iters, unsigned seed) {
  const int lane = threadIdx.x & 63;
  bf16x8 a[4], b[8];
  for (int q = 0; q < 8; ++q)
    for (int i = 0; i < 8; ++i) {
      unsigned hh = (lane * 8 + i + q * 977) * 2654435761u ^ seed; hh ^= hh >> 15; hh *= 2246822519u;
      b[q][i] = (__bf16)(((int)(hh >> 16) - 32768) / 32768.f);
      if (q < 4) a[q][i] = (__bf16)(((int)(hh & 0xffff) - 32768) / 32768.f);
    }
  f32x4 acc[4][8];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  v2f x[8], y[8]; unsigned h[8];
  for (int w = 0; w < 8; ++w) { x[w] = (v2f){lane * 0.01f + w, lane * 0.02f - w}; y[w] = x[w]; h[w] = lane * 77u + w; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (MF) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      if (VA) {
#pragma unroll
        for (int u = 0; u < UNITS; ++u) unit(x[(j + u) & 7], y[(j + u) & 7], h[(j + u) & 7]);
      }
      if (MF && VA) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x002 | 0x400, UNITS * 9, 0);   // VALU | TRANS
      }
    }
  }
  float sink = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) sink += acc[i][j][0];
  for (int w = 0; w < 8; ++w) sink += y[w].x + y[w].y + x[w].y;
  if (sink == 12345.678f) out[0] = sink;
}

template <int UNITS> void run(float *d, int iters) {
  auto time = [&](auto kern) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(s);
      hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, d, iters, 7u + rep);
      hipEventRecord(e); hipEventSynchronize(e);
      float ms; hipEventElapsedTime(&ms, s, e); if (rep && ms < best) best = ms;
    }
    return best * 1e3f;
  };
  const float tm = time(k<UNITS, true, false>), tv = time(k<UNITS, false, true>), tb = time(k<UNITS, true, true>);
  // cycles per trip at a nominal 2.4 GHz are not meaningful under DVFS: report times and the ratio to the sum
  printf("VALU per 32 MFMAs %4d: MFMAs alone %8.1f us  VALU alone %8.1f us  interleaved %8.1f us   (sum %8.1f, max %8.1f; interleaved / sum %.3f)\n",
         UNITS * 8 * 8, tm, tv, tb, tm + tv, tm > tv ? tm : tv, tb / (tm + tv));
}

int main() {
  float *d; hipMalloc(&d, 4);
  const int iters = 20000;
  // warm the clocks
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<1, true, true>), dim3(256), dim3(256), 0, 0, d, iters, 1u);
  hipDeviceSynchronize();
  run<1>(d, iters); run<2>(d, iters); run<3>(d, iters); run<4>(d, iters); run<6>(d, iters);
  return 0;
}
