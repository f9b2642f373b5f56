// Probe: the expert-MLP grouped GEMMs of the bench's layer shape (8 experts, rows = 8 * 1.25 * B*4096/8, H=704, I=2816) through
// the C ABI, cold caches (1 GiB read between launches), hipEvent timing.  The library source is compiled in, so -D
// switches of grouped_gemm.hip can be A/B'd as separate binaries inside one gpurun call.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-D...] tools/probes/gemm_probe.hip -o tools/probes/gemm_probe.bin
// run:   gemm_probe.bin [B=32] [H=704] [I=2816]     (config 3: 16 256 1024)
//
// Measured with it in round 2 (B=32, cold caches, same box per comparison; fc1 forward = [163840, 704] x 8 x [2816, 704]):
//   plain 806 us | + pre-activation output 942 | GELU only 890 | GELU + dropout 966 | GELU + pre 1000 | all three 1254;
//   fc2 data gradient with act'(pre)*mask 1073-1154; fc2 forward (N=704, K=2816, persistent 256^2 kernel) 617-654.
//   (a) wave priorities of the two-per-CU kernel (s_setprio 1 around the MFMA run on / off, epilogue at priority 0 / 1 / 2):
//       no difference (1234-1258 us): the epilogue is not losing an arbitration against the partner's K loop.
//   (b) epilogue straight from the accumulators - W rows permuted on their way into LDS so that a lane owns 16 consecutive
//       output columns (two 16-byte stores per row and output, no LDS staging pass, no barrier; commit 1a0d5e9 holds it):
//       correct (all GEMM tests green) and the same speed (1244 vs 1254; plain 811 vs 806): the staging pass is not what
//       the epilogue costs either.  Reverted.
//   (c) persistent 256^2 kernel, the LDS addresses of its DMA pieces: the solo path keeps eight of them in VGPRs and two spill
//       (44 B of scratch; each reload puts an `s_waitcnt vmcnt(1)` between two DMA instructions).  Built from a scalar base
//       the reloads leave the DMA path (20 B of scratch, all in the epilogue): 636-642 vs 626 us on fc2 forward, 800 vs 799
//       on the plain fc1 shape - not what bounds it.  The same DMA as inline asm with a memory clobber (the form of the
//       two-per-CU and TN kernels): 829 vs 613 us - the clobber pins the LDS reads and MFMAs around 64 DMA instructions per
//       step.  Both reverted.
//   (d) the SSM block's dense projections (`gemm_probe.bin 40 704 2816 dense`, one group, T = 163840 rows, cold caches), two-per-CU
//       kernel / persistent 256^2 kernel (-DNT_PROBE_FORCE=1 / 2) against the 5 TB/s streaming floor of operand + output bytes:
//       in_proj fwd (N=352, K=704) 119 / 133 us, floor 69 | x_param fwd (448, 176) 81 / 86, floor 41 | out_proj fwd (704, 176)
//       114 / 131, floor 58 | in_proj dgrad (704, 352) 150 / 177, floor 69 | x_param dgrad (176, 448) 60 / 67, floor 41 |
//       out_proj dgrad (176, 704) 85 / 87, floor 58: 54 % of the floor on the better kernel; L2 -> LDS fill (X re-read once per
//       128-wide n-tile) and the per-tile prologue / epilogue latency with two work-groups per CU, not HBM, bound them.
//   (e) config 3 (`gemm_probe.bin 16 256 1024`): DESIGN.md "Config 3".
//   (f) where the persistent kernel's time goes (`PROBE_SHORT=1 gemm_probe.bin 40`, -DNT_PROBE_FORCE=2 with -DNT_PROBE_EPI=1: no
//       epilogue, =2: epilogue without its global stores; B=40, cold caches, one box):
//         256 x 256 tile   fc1 shape (N=2816, K=704): 1044-1057 us | no stores 876 | no epilogue 738
//                          fc2 shape (N=704, K=2816):  818-849 us   | no stores 755 | no epilogue 708
//         256 x 352 tile   fc1 shape: 1050-1064                     | no stores 954 | no epilogue 670
//                          fc2 shape:  786-817                      | no stores 735 | no epilogue 690
//       i.e. on the 256-wide tile the exposed epilogue is 310 us (7.2 us per tile: 3.3 staging + barriers, 4.0 store issue) of the
//       short-K shape and 120 us of the long-K shape; a 128 KiB tile's stores issue at ~32 GB/s per CU whoever else is storing:
//       starting the work-groups of an XCD in 2 / 4 / 8 phases of a tile period (a spin on the wall clock before the first tile)
//       changed nothing (1056 / 1049 / 1085 us, 862 / 867 / 890) - it is not a chip-wide burst.  The K loops alone run at 53-60 %
//       of the MFMA peak: 64 (76) KiB of LDS-DMA per 1.6 (2.0) us step and CU = ~10 TB/s chip-wide on either tile - the 352-wide
//       tile moves 17 % fewer bytes per output and its K loop is 2.6 % (fc2) to 9 % (fc1 shape) shorter, not 17 %: the fill RATE a CU
//       reaches with one stage in flight (between the guide's 8.6 TB/s from the Infinity Cache and 17-19 from L2), not the byte
//       count, sets the step.  The two-per-CU kernel has 96 KiB in flight per CU and reaches 11.7 TB/s on 1.5x the bytes.
//       Tile order on the 352-wide kernel, m fastest inside an XCD's 32 tiles (one W panel per XCD, X shared through the Infinity
//       Cache) against n fastest (X shared in L2): 814-819 vs 798-802 us (fc2), 1106 vs 1053 (fc1 shape) - n fastest kept.
//   (g) what the K steps cost without their fills (-DNT_PROBE_EPI=1 -DNT_PROBE_NODMA: the K loop on stale LDS contents): fc2 shape
//       529 us (256-wide; 699 with the fills) and 496 us (352-wide; 683) - LDS reads, MFMAs and the barrier alone are 61-65 % of
//       the MFMA peak, the fills add 0.4-0.5 us to every 64-deep step although they are asynchronous: a stage issued at the top of
//       step k must have landed by the top of step k+1.  Tried on that evidence: the 352-wide kernel with a four-slot ring of
//       32-deep stages (38 KiB each, stage s+3 issued at the top of sub-step s: 114 KiB in flight per CU instead of 76; LDS image,
//       16-row x 64-byte pieces and in-order `vmcnt` bookkeeping of the two-per-CU kernel; correct in all tests).  SLOWER:
//       1278-1304 us against 786-817 (fc2 shape), 1353-1382 against 1050 (fc1 shape); without its fills 651 us (the doubled
//       barriers cost ~60 us) - the fills themselves are what is slow in that form: pieces of 16 rows x 64 B (half a cache line
//       per row and request) fill at about half the rate of 8 rows x 128 B from these operands.  A 64-deep stage cannot be
//       triple-buffered in 160 KiB at this tile size (3 x 76 KiB).  Removed.
#include "../../apertis_llm_amd/csrc/grouped_gemm.hip"
#include <cstdio>
#include <cstring>
#include <string>
#include <cstdlib>
#include <vector>
#include <algorithm>
__global__ void flush_read_k(const uint4 *p, size_t n, unsigned *sink) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc.x ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc.x == 0x12345678u) *sink = acc.x;
}
__global__ void fill_k(bf16_t *p, size_t n, float scale, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (bf16_t)(((int)(h & 0xffff) - 32768) * (scale / 32768.f));
  }
}
__global__ void checksum_k(const uint32_t *p, size_t n, unsigned long long *out) {
  unsigned long long a = 0, b = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { a += p[i]; b ^= (unsigned long long)p[i] * (i | 1); }
  atomicAdd(out, a); atomicXor(out + 1, b);
}
int main(int argc, char **argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32;
  const int64_t E = 8, H = argc > 2 ? atoi(argv[2]) : 704, I = argc > 3 ? atoi(argv[3]) : 2816, rows = E * (int64_t)((B * 4096 / E) * 1.25);
  bf16_t *x, *w1, *w2t, *h, *pre, *y, *dpre; float *b1; int32_t *offs; char *flush; unsigned *sink;
  hipMalloc(&x, rows * H * 2); hipMalloc(&w1, E * I * H * 2); hipMalloc(&w2t, E * I * H * 2); hipMalloc(&h, rows * I * 2);
  hipMalloc(&pre, rows * I * 2); hipMalloc(&y, rows * H * 2); hipMalloc(&dpre, rows * I * 2); hipMalloc(&b1, E * I * 4);
  hipMalloc(&offs, (E + 1) * 4); hipMalloc(&flush, (size_t)1 << 30); hipMalloc(&sink, 64);
  hipMemset(flush, 1, (size_t)1 << 30); hipMemset(b1, 0, E * I * 4);
  fill_k<<<2048, 256>>>(x, rows * H, 1.f, 1); fill_k<<<2048, 256>>>(w1, E * I * H, 0.05f, 2); fill_k<<<2048, 256>>>(w2t, E * I * H, 0.05f, 3);
  fill_k<<<2048, 256>>>(y, rows * H, 1.f, 4);
  std::vector<int32_t> ho(E + 1); for (int e = 0; e <= E; ++e) ho[e] = (int32_t)(rows * e / E);
  hipMemcpy(offs, ho.data(), (E + 1) * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char *name, double flops, auto fn) {
    float best = 1e9, sum = 0; int n = 0;
    const int reps = getenv("PROBE_REPS") ? atoi(getenv("PROBE_REPS")) : 6;
    for (int rep = 0; rep < reps; ++rep) {
      flush_read_k<<<4096, 256>>>((const uint4 *)flush, ((size_t)1 << 30) / 16, sink);
      hipEventRecord(e0); int rc = fn(); hipEventRecord(e1); hipEventSynchronize(e1);
      if (rc) { printf("%s rc=%d\n", name, rc); return; }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) { best = std::min(best, ms); sum += ms; ++n; }
      if (getenv("PROBE_ALL")) printf("   rep %d: %.1f us\n", rep, ms * 1e3);
    }
    printf("%-46s best %7.1f us  avg %7.1f us  %6.0f TF (best)\n", name, best * 1e3, sum / n * 1e3, flops / best / 1e9);
    if (getenv("PROBE_SUM")) {   // checksums of the three output buffers (two builds agree bit for bit <=> the lines agree)
      unsigned long long *cs; hipMalloc(&cs, 48); hipMemset(cs, 0, 48);
      checksum_k<<<1024, 256>>>((const uint32_t *)h, (size_t)rows * I / 2, cs);
      checksum_k<<<1024, 256>>>((const uint32_t *)pre, (size_t)rows * I / 2, cs + 2);
      checksum_k<<<1024, 256>>>((const uint32_t *)dpre, (size_t)rows * I / 2, cs + 4);
      unsigned long long hc[6]; hipMemcpy(hc, cs, 48, hipMemcpyDeviceToHost); hipFree(cs);
      printf("   sums h %016llx %016llx | pre %016llx %016llx | dpre %016llx %016llx\n", hc[0], hc[1], hc[2], hc[3], hc[4], hc[5]);
    }
#ifdef NT_PROBE_STAMPS
    unsigned long long st[8];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(nt2x_stamps), sizeof st);
    if (st[3]) printf("   two-per-CU work-groups: %llu, per work-group: prologue %.2f us, K loop %.2f us, epilogue %.2f us, own stores acknowledged after %.2f us\n",
                      st[3] / reps, st[0] * 0.01 / st[3], st[1] * 0.01 / st[3], st[2] * 0.01 / st[3], st[4] * 0.01 / st[3]);
    memset(st, 0, sizeof st);
    hipMemcpyToSymbol(HIP_SYMBOL(nt2x_stamps), st, sizeof st);
    hipMemcpyFromSymbol(st, HIP_SYMBOL(nt4r_stamps), sizeof st);
    if (st[2]) printf("   four-slot-ring tiles: %llu, per tile: K loop %.2f us, epilogue %.2f us; waits + barriers: sub-steps 0-2 %.2f us, sub-step 3 %.2f us, the rest %.2f us\n",
                      st[2] / reps, st[0] * 0.01 / st[2], st[1] * 0.01 / st[2], st[3] * 0.01 / st[2], st[4] * 0.01 / st[2], st[5] * 0.01 / st[2]);
    memset(st, 0, sizeof st);
    hipMemcpyToSymbol(HIP_SYMBOL(nt4r_stamps), st, sizeof st);
#endif
  };
  if (argc > 4) {   // dense projections of the SSM block (one group): gemm_probe.bin B 704 2816 dense
    const int64_t T = (int64_t)B * 4096;
    const int32_t o1[2] = {0, (int32_t)T};
    hipMemcpy(offs, o1, 8, hipMemcpyHostToDevice);
    const int shapes[6][2] = {{352, 704}, {448, 176}, {704, 176}, {704, 352}, {176, 448}, {176, 704}};   // (N, K)
    const char *names[6] = {"in_proj fwd   N=352 K=704", "x_param fwd   N=448 K=176", "out_proj fwd  N=704 K=176",
                            "in_proj dgrad N=704 K=352", "x_param dgrad N=176 K=448", "out_proj dgrad N=176 K=704"};
    for (int i = 0; i < 6; ++i) {
      const int64_t N = shapes[i][0], K = shapes[i][1], ldw = (K + 63) / 64 * 64;
      char nm[128]; snprintf(nm, sizeof nm, "%s  floor %5.1f us @5TB/s", names[i], T * (N + K) * 2 / 5e6);
      timeit(nm, 2.0 * T * N * K, [&] { return apertis_grouped_gemm_nt(h, w1, nullptr, offs, pre, nullptr, nullptr, T, N, K, ldw, 1, APERTIS_ACT_NONE, 0.f, 0, APERTIS_BF16, APERTIS_BF16, nullptr); });
    }
    return 0;
  }
  const double fl = 2.0 * rows * H * I;
  if (getenv("PROBE_R3")) {   // round 3 (build with -DNT_PROBE_WALK -DTN_PROBE_RING): tile walks of the two-per-CU kernel, TN ring
    float *dw1, *dw2, *db1, *db2, *ws;
    hipMalloc(&dw1, E * I * H * 4); hipMalloc(&dw2, E * I * H * 4); hipMalloc(&db1, E * I * 4); hipMalloc(&db2, E * H * 4);
    const int64_t wsb = apertis_grouped_gemm_tn_workspace_bytes(E, 2);
    hipMalloc(&ws, wsb);
    fill_k<<<2048, 256>>>(h, rows * I, 1.f, 5); fill_k<<<2048, 256>>>(dpre, rows * I, 1.f, 6);
    const char *only = getenv("PROBE_ONLY");
    if (!only || !strcmp(only, "walk")) {
      std::vector<std::string> walks = {"0,0", "16,11", "32,4", "16,4", "8,11", "32,11", "24,11", "8,4", "64,4", "16,3", "16,2", "32,2", "12,11", "16,11"};
      if (const char *wl = getenv("PROBE_WALKS")) {   // e.g. PROBE_WALKS="0,0;16,11"
        walks.clear();
        std::string t(wl); size_t a = 0;
        while (a <= t.size()) { size_t b = t.find(';', a); if (b == std::string::npos) b = t.size(); if (b > a) walks.push_back(t.substr(a, b - a)); a = b + 1; }
      }
      for (const std::string &wks : walks) {
        const char *wk = wks.c_str();
        setenv("NT_WALK", wk, 1);
        char nm[3][96];
        snprintf(nm[0], 96, "walk %-8s fc1 fwd GELU+drop+g' (SAVE_GRAD)", wk);
        snprintf(nm[1], 96, "walk %-8s fc2 dgrad * saved g' (MUL_SAVED)", wk);
        snprintf(nm[2], 96, "walk %-8s fc1 shape plain", wk);
        timeit(nm[0], fl, [&] { return apertis_grouped_gemm_nt(x, w1, b1, offs, h, pre, nullptr, rows, I, H, H, E, APERTIS_ACT_GELU | APERTIS_ACT_SAVE_GRAD, 0.1f, 7, APERTIS_BF16, APERTIS_BF16, nullptr); });
        timeit(nm[1], fl, [&] { return apertis_grouped_gemm_nt(y, w2t, nullptr, offs, dpre, nullptr, pre, rows, I, H, H, E, APERTIS_ACT_MUL_SAVED, 0.f, 0, APERTIS_BF16, APERTIS_BF16, nullptr); });
        timeit(nm[2], fl, [&] { return apertis_grouped_gemm_nt(x, w1, b1, offs, h, nullptr, nullptr, rows, I, H, H, E, APERTIS_ACT_NONE, 0.f, 0, APERTIS_BF16, APERTIS_BF16, nullptr); });
      }
    }
    if (!only || !strcmp(only, "tn")) {
      for (const char *rg : {"0", "2", "5", "2", "5"}) {   // 0 / 2: v3 double buffer / ring + stagger; 5: v5 (256 x 352 tiles)
        setenv("TN_V5", rg[0] == '5' ? "1" : "0", 1);
        setenv("TN_RING", rg[0] == '5' ? "2" : rg, 1);
        char nm[96]; snprintf(nm, 96, "TN pair (dW2 = dy^T h, dW1 = dpre^T x), kernel=%s", rg);
        timeit(nm, 2 * fl, [&] { return apertis_grouped_gemm_tn_pair(y, h, dw2, db2, H, I, dpre, x, dw1, db1, I, H, offs, rows, E, ws, wsb, APERTIS_BF16, nullptr); });
        snprintf(nm, 96, "TN as two launches (32 CUs per expert), ring=%s", rg);
        timeit(nm, 2 * fl, [&] {
          int rc = apertis_grouped_gemm_tn(y, h, offs, dw2, db2, rows, H, I, E, ws, wsb, APERTIS_BF16, nullptr);
          return rc ? rc : apertis_grouped_gemm_tn(dpre, x, offs, dw1, db1, rows, I, H, E, ws, wsb, APERTIS_BF16, nullptr); });
      }
    }
    return 0;
  }
  if (getenv("PROBE_R4")) {   // round 4: the four calls that matter for the expert path at K = H
    fill_k<<<2048, 256>>>(pre, rows * I, 1.f, 9);
    timeit("fc1 fwd: GELU + dropout + g' out (SAVE_GRAD)", fl, [&] { return apertis_grouped_gemm_nt(x, w1, b1, offs, h, pre, nullptr, rows, I, H, H, E, APERTIS_ACT_GELU | APERTIS_ACT_SAVE_GRAD, 0.1f, 7, APERTIS_BF16, APERTIS_BF16, nullptr); });
    timeit("fc2 dgrad * saved g' (MUL_SAVED)", fl, [&] { return apertis_grouped_gemm_nt(y, w2t, nullptr, offs, dpre, nullptr, pre, rows, I, H, H, E, APERTIS_ACT_MUL_SAVED, 0.f, 0, APERTIS_BF16, APERTIS_BF16, nullptr); });
    timeit("fc1 fwd shape, plain (one output)", fl, [&] { return apertis_grouped_gemm_nt(x, w1, b1, offs, h, nullptr, nullptr, rows, I, H, H, E, APERTIS_ACT_NONE, 0.f, 0, APERTIS_BF16, APERTIS_BF16, nullptr); });
    timeit("fc2 fwd (N=H, K=I), plain", fl, [&] { return apertis_grouped_gemm_nt(h, w2t, nullptr, offs, y, nullptr, nullptr, rows, H, I, I, E, APERTIS_ACT_NONE, 0.f, 0, APERTIS_BF16, APERTIS_BF16, nullptr); });
    return 0;
  }
  if (getenv("PROBE_SHORT")) {   // the two plain shapes only (epilogue probes: -DNT_PROBE_FORCE=2 -DNT_PROBE_EPI=1|2)
    timeit("fc1 fwd shape (N=I, K=H), plain (one output)", fl, [&] { return apertis_grouped_gemm_nt(x, w1, b1, offs, h, nullptr, nullptr, rows, I, H, H, E, APERTIS_ACT_NONE, 0.f, 0, APERTIS_BF16, APERTIS_BF16, nullptr); });
    timeit("fc2 fwd (N=H, K=I), plain", fl, [&] { return apertis_grouped_gemm_nt(h, w2t, nullptr, offs, y, nullptr, nullptr, rows, H, I, I, E, APERTIS_ACT_NONE, 0.f, 0, APERTIS_BF16, APERTIS_BF16, nullptr); });
    return 0;
  }
  timeit("fc1 fwd: GELU + dropout + pre-activation out", fl, [&] { return apertis_grouped_gemm_nt(x, w1, b1, offs, h, pre, nullptr, rows, I, H, H, E, APERTIS_ACT_GELU, 0.1f, 7, APERTIS_BF16, APERTIS_BF16, nullptr); });
  timeit("fc1 fwd shape, plain (one output)", fl, [&] { return apertis_grouped_gemm_nt(x, w1, b1, offs, h, nullptr, nullptr, rows, I, H, H, E, APERTIS_ACT_NONE, 0.f, 0, APERTIS_BF16, APERTIS_BF16, nullptr); });
  timeit("fc1 shape: + pre-activation out only", fl, [&] { return apertis_grouped_gemm_nt(x, w1, b1, offs, h, pre, nullptr, rows, I, H, H, E, APERTIS_ACT_NONE, 0.f, 0, APERTIS_BF16, APERTIS_BF16, nullptr); });
  timeit("fc1 shape: GELU only (one output)", fl, [&] { return apertis_grouped_gemm_nt(x, w1, b1, offs, h, nullptr, nullptr, rows, I, H, H, E, APERTIS_ACT_GELU, 0.f, 0, APERTIS_BF16, APERTIS_BF16, nullptr); });
  timeit("fc1 shape: GELU + dropout (one output)", fl, [&] { return apertis_grouped_gemm_nt(x, w1, b1, offs, h, nullptr, nullptr, rows, I, H, H, E, APERTIS_ACT_GELU, 0.1f, 7, APERTIS_BF16, APERTIS_BF16, nullptr); });
  timeit("fc1 shape: GELU + pre out (no dropout)", fl, [&] { return apertis_grouped_gemm_nt(x, w1, b1, offs, h, pre, nullptr, rows, I, H, H, E, APERTIS_ACT_GELU, 0.f, 0, APERTIS_BF16, APERTIS_BF16, nullptr); });
  timeit("fc2 dgrad * act'(pre) * mask", fl, [&] { return apertis_grouped_gemm_nt(y, w2t, nullptr, offs, dpre, nullptr, pre, rows, I, H, H, E, APERTIS_ACT_GELU, 0.1f, 7, APERTIS_BF16, APERTIS_BF16, nullptr); });
  timeit("fc1 fwd: GELU + dropout + g' out (SAVE_GRAD)", fl, [&] { return apertis_grouped_gemm_nt(x, w1, b1, offs, h, pre, nullptr, rows, I, H, H, E, APERTIS_ACT_GELU | APERTIS_ACT_SAVE_GRAD, 0.1f, 7, APERTIS_BF16, APERTIS_BF16, nullptr); });
  timeit("fc1 fwd: GELU + g' out, no dropout (SAVE_GRAD)", fl, [&] { return apertis_grouped_gemm_nt(x, w1, b1, offs, h, pre, nullptr, rows, I, H, H, E, APERTIS_ACT_GELU | APERTIS_ACT_SAVE_GRAD, 0.f, 0, APERTIS_BF16, APERTIS_BF16, nullptr); });
  timeit("fc2 dgrad * saved g' (MUL_SAVED)", fl, [&] { return apertis_grouped_gemm_nt(y, w2t, nullptr, offs, dpre, nullptr, pre, rows, I, H, H, E, APERTIS_ACT_MUL_SAVED, 0.f, 0, APERTIS_BF16, APERTIS_BF16, nullptr); });
  timeit("fc2 fwd (N=H, K=I), plain", fl, [&] { return apertis_grouped_gemm_nt(h, w2t, nullptr, offs, y, nullptr, nullptr, rows, H, I, I, E, APERTIS_ACT_NONE, 0.f, 0, APERTIS_BF16, APERTIS_BF16, nullptr); });
  return 0;
}
