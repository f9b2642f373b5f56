// Probe: what a concurrent kernel that HOLDS some CUs (an RCCL collective on the communication stream of the data-parallel
// step) does to the one-work-group-per-CU persistent GEMMs - measured on ONE GPU with a synthetic CU hog instead of
// estimated.  The hog is `nhog` work-groups of 1024 threads with the CU's whole LDS each, spinning on
// the wall clock for `hog_us`; a GEMM work-group (128+ KiB of LDS, a full register file) cannot share a CU with it.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/probes/hog_probe.hip -o tools/probes/hog_probe.bin
// run:   hog_probe.bin [B=32] [nhog=32] [hog_us=4000]
// Also measured with it (B=40, no hog): the weight-gradient kernel with 32 work-groups per (problem, expert) group instead of the
// CUs' equal share of 16 (one round of 32 tiles + one tile split 32 ways, the grid in two dispatch waves; per K step 16
// operand panels instead of 20 would reach L2 from HBM): 1928-1939 us against 1771-1789 - slower, not kept.
#include "../../apertis_llm_amd/csrc/grouped_gemm.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
#include <unistd.h>

__global__ void __launch_bounds__(1024) hog_k(long long ticks, unsigned *sink) {
  extern __shared__ char hog_lds[];
  const long long t0 = wall_clock64();
  unsigned acc = 0;
  while (wall_clock64() - t0 < ticks) { acc += (unsigned)hog_lds[(threadIdx.x * 64) & 0xffff]; __builtin_amdgcn_s_sleep(8); }
  if (acc == 0x12345678u) *sink = acc;
}
__global__ void fill_k(bf16_t *p, size_t n, float scale, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (bf16_t)(((int)(h & 0xffff) - 32768) * (scale / 32768.f));
  }
}

int main(int argc, char **argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32, nhog = argc > 2 ? atoi(argv[2]) : 32;
  const double hog_us = argc > 3 ? atof(argv[3]) : 4000.0;
  const int64_t E = 8, H = 704, I = 2816, rows = E * (int64_t)((B * 4096 / E) * 1.25);
  bf16_t *x, *w2t, *h, *y, *dyr, *dpre; float *dw1, *dw2, *db1, *db2; int32_t *offs, *queue; unsigned *sink; void *ws;
  hipMalloc(&x, rows * H * 2); hipMalloc(&w2t, E * I * H * 2); hipMalloc(&h, rows * I * 2); hipMalloc(&y, rows * H * 2);
  hipMalloc(&dyr, rows * H * 2); hipMalloc(&dpre, rows * I * 2); hipMalloc(&dw1, E * I * H * 4); hipMalloc(&dw2, E * I * H * 4);
  hipMalloc(&db1, E * I * 4); hipMalloc(&db2, E * H * 4); hipMalloc(&offs, (E + 1) * 4); hipMalloc(&queue, 4 * APERTIS_NT_QUEUE_INTS); hipMalloc(&sink, 64);
  const int64_t wsb = apertis_grouped_gemm_tn_workspace_bytes(E, 2);
  hipMalloc(&ws, wsb);
  fill_k<<<2048, 256>>>(x, rows * H, 1.f, 1); fill_k<<<2048, 256>>>(w2t, E * I * H, 0.05f, 3); fill_k<<<2048, 256>>>(h, rows * I, 1.f, 5);
  fill_k<<<2048, 256>>>(dyr, rows * H, 1.f, 6); fill_k<<<2048, 256>>>(dpre, rows * I, 1.f, 7);
  std::vector<int32_t> ho(E + 1); for (int e = 0; e <= E; ++e) ho[e] = (int32_t)(rows * e / E);
  hipMemcpy(offs, ho.data(), (E + 1) * 4, hipMemcpyHostToDevice);
  hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  hipFuncSetAttribute((const void *)hog_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char *name, bool hog, auto fn) {
    float best = 1e9, sum = 0; int n = 0;
    for (int rep = 0; rep < 5; ++rep) {
      hipDeviceSynchronize();
      if (hog) {
        hipLaunchKernelGGL(hog_k, dim3(nhog), dim3(1024), 160 * 1024, sb, (long long)(hog_us * 100.0), sink);   // 100 MHz wall clock
        usleep(300);   // the hog is resident before the GEMM is queued
      }
      hipEventRecord(e0, sa); int rc = fn(sa); hipEventRecord(e1, sa); hipEventSynchronize(e1);
      if (rc) { printf("%s rc=%d\n", name, rc); return; }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) { best = std::min(best, ms); sum += ms; ++n; }
    }
    printf("%-58s %-9s best %7.1f us  avg %7.1f us\n", name, hog ? "with hog" : "alone", best * 1e3, sum / n * 1e3);
  };
  printf("B=%d rows=%ld; hog = %d work-groups (one CU each) for %.0f us\n", B, (long)rows, nhog, hog_us);
  for (int hog = 0; hog < 2; ++hog) {
    timeit("NT persistent 256x352 (fc2 fwd N=704 K=2816), static walk", hog, [&](hipStream_t s) {
      return apertis_grouped_gemm_nt_q(h, w2t, nullptr, offs, y, nullptr, nullptr, rows, H, I, I, E, APERTIS_ACT_NONE, 0.f, 0, APERTIS_BF16, APERTIS_BF16, nullptr, s); });
    timeit("NT persistent 256x352 (fc2 fwd N=704 K=2816), tile queue", hog, [&](hipStream_t s) {
      return apertis_grouped_gemm_nt_q(h, w2t, nullptr, offs, y, nullptr, nullptr, rows, H, I, I, E, APERTIS_ACT_NONE, 0.f, 0, APERTIS_BF16, APERTIS_BF16, queue, s); });
    timeit("NT two-per-CU (fc1 fwd N=2816 K=704, GELU+dropout+pre)", hog, [&](hipStream_t s) {
      return apertis_grouped_gemm_nt_q(x, w2t, nullptr, offs, h, dpre, nullptr, rows, I, H, H, E, APERTIS_ACT_GELU, 0.1f, 7, APERTIS_BF16, APERTIS_BF16, nullptr, s); });
    timeit("TN pair 256x256 (dW1, dW2 of the expert MLP), static walk", hog, [&](hipStream_t s) {
      return apertis_grouped_gemm_tn_pair_q(dpre, x, dw1, db1, I, H, dyr, h, dw2, db2, H, I, offs, rows, E, ws, wsb, APERTIS_BF16, 0, s); });
    timeit("TN pair 256x256 (dW1, dW2 of the expert MLP), item queue", hog, [&](hipStream_t s) {
      return apertis_grouped_gemm_tn_pair_q(dpre, x, dw1, db1, I, H, dyr, h, dw2, db2, H, I, offs, rows, E, ws, wsb, APERTIS_BF16, 1, s); });
  }
  return 0;
}
