// Probe: L2 -> LDS fill rate of the NT GEMM's operand pattern (tools/probes/dma_fill.hip reads contiguous KiB):
// each WG owns tile (mt, nt) and per K step DMAs 256 rows x 128 B of X (row pitch ldx) and of W (pitch ldw) at
// column offset step*128.  Tiles are laid out over XCDs as grouped_gemm_nt256p_k does (32 consecutive tiles per XCD).
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/dma_tile.hip -o /tmp/dma_tile
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(const v4i &rs, uint32_t lds, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(__builtin_amdgcn_readfirstlane((int)lds)), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ v4i rsrc(const void *p, uint32_t bytes) {
  const uint64_t a = (uint64_t)p;
  v4i rs; rs[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a); rs[1] = __builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffff));
  rs[2] = (int)bytes; rs[3] = 0x00020000; return rs;
}
// inflight = stages in flight (1: wait all, 2: leave 8 per wave outstanding)
__global__ void __launch_bounds__(512) tile_k(const char *X, const char *W, int ldx, int ldw, int nk, int n_tiles, int rounds, int inflight, int swz) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char *)smem;
  const int G = gridDim.x;
  for (int r = 0; r < rounds; ++r) {
    const int within = blockIdx.x;
    const int tile = r * G + (within % 8) * (G / 8) + within / 8;
    const int mt = tile / n_tiles, nt = tile % n_tiles;
    const v4i xrs = rsrc(X + (size_t)mt * 256 * ldx, 256u * ldx), wrs = rsrc(W + (size_t)nt * 256 * ldw, 256u * ldw);
    const int chunk = swz ? ((lane & 7) ^ (lane >> 3)) << 4 : (lane & 7) << 4;
    const uint32_t vx = (lane >> 3) * ldx + chunk, vw = (lane >> 3) * ldw + chunk;
    for (int kt = 0; kt < nk; ++kt) {
      char buf = kt & 1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int p = wave * 4 + j;
        dma16(xrs, lds0 + buf * 65536 + p * 1024, vx + p * 8 * ldx, kt * 128);
        dma16(wrs, lds0 + buf * 65536 + 32768 + p * 1024, vw + p * 8 * ldw, kt * 128);
      }
      if (inflight == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __syncthreads();
    }
  }
}
int main(int argc, char **argv) {
  const int M_T = argc > 1 ? atoi(argv[1]) : 64, N_T = 16;   // M_T m-tiles x 16 n-tiles, rounds of 256 tiles
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void *)tile_k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  const int Ks[] = {704, 2816, 4096};
  for (int ki = 0; ki < 3; ++ki)
    for (int inflight = 1; inflight <= 2; ++inflight)
      for (int swz = 1; swz < 2; ++swz) {
        const int ld = Ks[ki] * 2, nk = (Ks[ki] & ~63) / 64;
        char *X, *W;
        hipMalloc(&X, (size_t)M_T * 256 * ld + 4096); hipMalloc(&W, (size_t)N_T * 256 * ld + 4096);
        hipMemset(X, 1, (size_t)M_T * 256 * ld); hipMemset(W, 1, (size_t)N_T * 256 * ld);
        hipLaunchKernelGGL(tile_k, dim3(256), dim3(512), 131072, 0, X, W, ld, ld, nk, N_T, M_T / 16, inflight, swz);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(tile_k, dim3(256), dim3(512), 131072, 0, X, W, ld, ld, nk, N_T, M_T / 16, inflight, swz);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = 256.0 * (M_T / 16) * nk * 65536;
        printf("ld=%5d B nk=%2d inflight=%d swz=%d: %8.1f us  %6.2f TB/s  (GEMM-equivalent %5.0f TF)\n", ld, nk, inflight, swz, ms * 1e3,
               bytes / ms / 1e9, bytes * 128 / ms / 1e9 / 1e3);
        hipFree(X); hipFree(W);
      }
  return 0;
}
