// Probe build of the fused scan's forward: the library source compiled with -DSCAN_PROBE so that every work-group records
// wall-clock timestamps at its phase boundaries; launched through the C ABI at the bench's per-layer shape.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSCAN_PROBE tools/probes/scan_gate_probe.hip -o tools/probes/scan_gate_probe.bin
#include "../../apertis_llm_amd/csrc/scan_gate.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
// evict the caches with READS of 1 GiB (clean lines: nothing left to write back under the timed kernel)
__global__ void flush_read_k(const uint4 *p, size_t n, unsigned *sink) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc.x ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc.x == 0x12345678u) *sink = acc.x;
}
static unsigned *g_sink = nullptr;
static void flush_read(const char *buf) {
  if (!g_sink) hipMalloc(&g_sink, 64);
  hipLaunchKernelGGL(flush_read_k, dim3(4096), dim3(256), 0, 0, (const uint4 *)buf, ((size_t)1 << 30) / 16, g_sink);
}
int main(int argc, char **argv) {
  const int64_t B = argc > 1 ? atoi(argv[1]) : 32, L = 4096, h = 11, N = 16, Dn = h * N;
  const int single = argc > 2 ? atoi(argv[2]) : 1;
  const int64_t T = B * L;
  char *p, *xz, *xc, *out; float *dlt, *A, *D, *h_in, *agg; void *ws;
  hipMalloc(&p, T * 896); hipMalloc(&xz, T * 704); hipMalloc(&xc, T * 352); hipMalloc(&out, T * 352);
  hipMalloc(&dlt, T * h * 4); hipMalloc(&A, h * N * 4); hipMalloc(&D, Dn * 4); hipMalloc(&h_in, B * 64 * Dn * 4);
  hipMalloc(&agg, B * 64 * Dn * 8);
  const int64_t wsb = apertis_scan_gate_workspace_bytes(B, L, Dn);
  hipMalloc(&ws, wsb); hipMemset(ws, 0, wsb);
  hipMemset(p, 0x3c, T * 896); hipMemset(xz, 0x3c, T * 704); hipMemset(xc, 0x3c, T * 352);
  std::vector<float> hd(T * h, -4.f), ha(h * N, -0.3f), hD(Dn, 1.f);
  hipMemcpy(dlt, hd.data(), T * h * 4, hipMemcpyHostToDevice); hipMemcpy(A, ha.data(), h * N * 4, hipMemcpyHostToDevice);
  hipMemcpy(D, hD.data(), Dn * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  uint32_t epoch = 0;
  float best = 1e9;
  char *flush; hipMalloc(&flush, (size_t)1 << 30); hipMemset(flush, 1, (size_t)1 << 30); hipDeviceSynchronize();
  for (int rep = 0; rep < 5; ++rep) {
    flush_read(flush);
    hipEventRecord(e0);
    int rc = apertis_scan_gate_fwd(dlt, A, p, 448, p + 384, 448, xc, 176, xz + 352, 352, D, nullptr, out, 176, nullptr, agg, h_in, ws,
                                   ++epoch, B, L, h, N, APERTIS_BF16, 1, single, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rc) { printf("rc=%d\n", rc); return 1; }
    if (rep) best = std::min(best, ms);
  }
  const int do_bwd = argc > 3 ? atoi(argv[3]) : 0;
  if (do_bwd) {
    char *dout, *dp, *dxz, *dxc; float *ddl, *dAD, *fold, *part;
    hipMalloc(&dout, T * 352); hipMalloc(&dp, T * 896); hipMalloc(&dxz, T * 704); hipMalloc(&dxc, T * 352);
    hipMalloc(&ddl, T * h * 4); hipMalloc(&dAD, 2 * Dn * 4); hipMalloc(&fold, 64 * 2 * Dn * 4); hipMalloc(&part, B * 64 * 2 * Dn * 4);
    hipMemset(dout, 0x3c, T * 352);
    best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      flush_read(flush);
      hipEventRecord(e0);
      int rc = apertis_scan_gate_bwd(dlt, A, p, 448, p + 384, 448, xc, 176, xz + 352, 352, D, dout, 176, h_in, dp, 448, dp + 384, 448, 192,
                                     dxc, 176, dxz + 352, 352, ddl, dAD, agg, fold, part, ws, ++epoch, B, L, h, N, APERTIS_BF16, 1, single,
                                     nullptr);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rc) { printf("bwd rc=%d\n", rc); return 1; }
      if (rep) best = std::min(best, ms);
    }
  }
  const int items = (int)(B * 64);
  std::vector<unsigned long long> pr(8192 * 8);
  hipMemcpyFromSymbol(pr.data(), HIP_SYMBOL(g_probe), pr.size() * 8);
  double ph[8] = {0}; unsigned long long tmin = ~0ull, tmax = 0;
  for (int i = 0; i < std::min(items, 8192); ++i) {
    for (int k = 1; k < 8; ++k) ph[k] += (double)(pr[i * 8 + k] - pr[i * 8 + k - 1]);
    tmin = std::min(tmin, pr[i * 8]); tmax = std::max(tmax, pr[i * 8 + 7]);
  }
  printf("%s single=%d B=%ld: %.1f us by events (%.2f TB/s, cold caches); wall_clock span %.1f us at 100 MHz\n", do_bwd ? "bwd" : "fwd", single,
         (long)B, best * 1e3, (double)T * ((do_bwd ? 9 : 5) * Dn * 2 + (do_bwd ? 8 : 4) * h) / best / 1e9, (double)(tmax - tmin) / 100.0);
  const char *names[8] = {"", "loads issued -> Bt landed + barrier", "segment aggregates + barrier", "publish + gather",
                          "C tile to LDS + barrier", "replay + barrier", "y tile + barrier", "epilogue + stores issued"};
  for (int k = 1; k < 8; ++k) printf("  phase %d %-40s avg %7.2f us per work-group\n", k, names[k], ph[k] / items / 100.0);
  // concurrency: average number of work-groups alive
  double alive = 0; for (int i = 0; i < std::min(items, 8192); ++i) alive += (double)(pr[i * 8 + 7] - pr[i * 8]);
  printf("  work-group lifetime avg %.2f us; average work-groups in flight %.0f\n", alive / items / 100.0, alive / (double)(tmax - tmin));
  return 0;
}
