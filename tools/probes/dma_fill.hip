// Probe: peak L2 -> LDS fill rate of buffer_load ... lds (LDS-DMA), all CUs, operand set resident in L2.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/dma_fill.hip -o /tmp/dma_fill ; run: /tmp/dma_fill
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(const v4i &rs, uint32_t lds, uint32_t voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(__builtin_amdgcn_readfirstlane((int)lds)), "v"(voff), "s"(rs) : "memory", "m0");
}
// each WG (nwaves*64 threads) repeatedly fills `bytes_per_step` of LDS from a per-XCD-shared region of `span` bytes
template <int NW>
__global__ void __launch_bounds__(NW * 64) fill_k(const char *src, uint32_t span, int steps, int pieces_per_wave, int wg_stride, float *sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t a = (uint64_t)src;
  v4i rs; rs[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a); rs[1] = __builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffff));
  rs[2] = (int)span; rs[3] = 0x00020000;
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char *)smem;
  uint32_t off = (uint32_t)((blockIdx.x % 8) * wg_stride);   // neighbours share lines (like tiles sharing operand blocks)
  for (int s = 0; s < steps; ++s) {
    for (int p = 0; p < pieces_per_wave; ++p) {
      const uint32_t piece = (uint32_t)((wave * pieces_per_wave + p) * 1024);
      dma16(rs, lds0 + (s & 1) * NW * pieces_per_wave * 1024 + piece, (off + piece + lane * 16) % span);
    }
    off += NW * pieces_per_wave * 1024;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (sink && threadIdx.x == 0) sink[blockIdx.x] = ((float *)smem)[0];
}
int main() {
  const size_t span = 2u << 20;   // 2 MiB: L2-resident per XCD
  char *src; float *sink;
  hipMalloc(&src, span); hipMemset(src, 1, span); hipMalloc(&sink, 4096 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int cfg = 0; cfg < 4; ++cfg) {
    const int nw = cfg < 2 ? 8 : 4, ppw = cfg == 0 ? 8 : cfg == 1 ? 8 : 6, wgs_per_cu = cfg == 0 ? 1 : cfg == 1 ? 1 : 2;
    const int steps = 4000;
    const size_t lds = (size_t)2 * nw * ppw * 1024;
    auto k = nw == 8 ? fill_k<8> : fill_k<4>;
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL(k, dim3(grid), dim3(nw * 64), lds, 0, src, (uint32_t)span, 10, ppw, 65536, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(nw * 64), lds, 0, src, (uint32_t)span, steps, ppw, 65536, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * steps * nw * ppw * 1024;
    printf("waves/WG %d pieces/wave/step %d WGs/CU %d (%zu KiB/step/WG): %.2f TB/s  (%.2f us/step)\n", nw, ppw, wgs_per_cu,
           (size_t)nw * ppw, bytes / ms / 1e9, ms * 1e3 / steps);
  }
  return 0;
}
