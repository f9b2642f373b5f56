// Probe: what HBM rate does the fused scan's ACCESS PATTERN allow, independent of its arithmetic?
// Work item = (batch b, chunk of LT tokens, 64-channel tile): four input tiles of LT rows x 128 B (bf16) at the model's row
// pitches (Bt and C inside p [T,448], xc [T,176], z inside xz [T,352]) and one output tile (out [T,176]).
//   mode 0: one work-group per item (ticket order = chunk-major, like scan_gate_fwd_k), all loads issued up front,
//           a __syncthreads(), then the store.  `lds` bytes of dynamic LDS pin the work-groups per CU.
//   mode 1: persistent work-groups (grid = 256 * per_cu) walk the items with the NEXT item's loads issued before the
//           current item's barrier + store (register double buffer).
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/scan_stream.hip -o tools/probes/scan_stream.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

static char *g_flush = nullptr;          // 1 GiB written before every timed launch: the Infinity Cache (256 MiB) starts cold
static void flush_caches() {
  if (!g_flush) hipMalloc(&g_flush, (size_t)1 << 30);
  hipMemsetAsync(g_flush, 0x5a, (size_t)1 << 30, 0);
}

struct Args {
  const char *p, *xc, *xz;
  char *out;
  int B, L, ctiles, nchunks, LT, items;
  int p_pitch, xc_pitch, xz_pitch, c_off, z_off;
};

template <int NTH, int LT>
struct Tiles {
  static constexpr int TOTAL = LT * 8, ITERS = (TOTAL + NTH - 1) / NTH;
  uint4 a[ITERS], b[ITERS], c[ITERS], d[ITERS];
  __device__ __forceinline__ void load(const Args &A, int item, int tid) {
    const int bct = A.B * A.ctiles, chunk = item / bct, r = item - chunk * bct, bb = r / A.ctiles, ct = r - bb * A.ctiles;
    const int64_t tok0 = (int64_t)bb * A.L + (int64_t)chunk * LT;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int idx = tid + it * NTH, row = idx >> 3, cb = (idx & 7) * 16;
      const bool ok = idx < TOTAL && ct * 128 + cb < 352;
      const uint4 zz = make_uint4(0, 0, 0, 0);
      a[it] = ok ? *reinterpret_cast<const uint4 *>(A.p + (tok0 + row) * A.p_pitch + ct * 128 + cb) : zz;
      b[it] = ok ? *reinterpret_cast<const uint4 *>(A.p + (tok0 + row) * A.p_pitch + A.c_off + ct * 128 + cb) : zz;
      c[it] = ok ? *reinterpret_cast<const uint4 *>(A.xc + (tok0 + row) * A.xc_pitch + ct * 128 + cb) : zz;
      d[it] = ok ? *reinterpret_cast<const uint4 *>(A.xz + (tok0 + row) * A.xz_pitch + A.z_off + ct * 128 + cb) : zz;
    }
  }
  __device__ __forceinline__ void store(const Args &A, int item, int tid) const {
    const int bct = A.B * A.ctiles, chunk = item / bct, r = item - chunk * bct, bb = r / A.ctiles, ct = r - bb * A.ctiles;
    const int64_t tok0 = (int64_t)bb * A.L + (int64_t)chunk * LT;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int idx = tid + it * NTH, row = idx >> 3, cb = (idx & 7) * 16;
      if (idx < TOTAL && ct * 128 + cb < 352) {
        uint4 o = make_uint4(a[it].x ^ b[it].x ^ c[it].x ^ d[it].x, a[it].y + b[it].y + c[it].y + d[it].y, a[it].z ^ d[it].z,
                             b[it].w ^ c[it].w);
        *reinterpret_cast<uint4 *>(A.out + (tok0 + row) * A.xc_pitch + ct * 128 + cb) = o;
      }
    }
  }
};

template <int NTH, int LT>
__global__ void __launch_bounds__(NTH) oneshot_k(Args A, unsigned *ctr) {
  extern __shared__ char smem[];
  __shared__ int item_s;
  if (threadIdx.x == 0) item_s = (int)atomicAdd(ctr, 1u);
  __syncthreads();
  const int item = item_s;
  Tiles<NTH, LT> t;
  t.load(A, item, threadIdx.x);
  if (smem[threadIdx.x & 15] == 77) t.a[0].x ^= 1;   // keep the dynamic LDS alive
  __syncthreads();
  t.store(A, item, threadIdx.x);
}

template <int NTH, int LT>
__global__ void __launch_bounds__(NTH) persistent_k(Args A, unsigned *ctr) {
  extern __shared__ char smem[];
  __shared__ int item_s[2];
  if (threadIdx.x == 0) item_s[0] = (int)atomicAdd(ctr, 1u);
  __syncthreads();
  int item = item_s[0], par = 0;
  Tiles<NTH, LT> cur, nxt;
  if (item < A.items) cur.load(A, item, threadIdx.x);
  while (item < A.items) {
    if (threadIdx.x == 0) item_s[par ^ 1] = (int)atomicAdd(ctr, 1u);
    __syncthreads();
    const int nitem = item_s[par ^ 1];
    if (nitem < A.items) nxt.load(A, nitem, threadIdx.x);
    if (smem[threadIdx.x & 15] == 77) cur.a[0].x ^= 1;
    cur.store(A, item, threadIdx.x);
    cur = nxt;
    item = nitem;
    par ^= 1;
  }
}

template <int NTH, int LT> void run(const char *name, int mode, Args A, unsigned *ctr, int lds, int per_cu) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  A.LT = LT; A.nchunks = A.L / LT; A.items = A.nchunks * A.B * A.ctiles;
  float best = 1e9;
  for (int rep = 0; rep < 6; ++rep) {
    flush_caches();
    hipMemsetAsync(ctr, 0, 4, 0);
    hipEventRecord(e0);
    if (mode == 0) {
      hipFuncSetAttribute((const void *)oneshot_k<NTH, LT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipLaunchKernelGGL((oneshot_k<NTH, LT>), dim3(A.items), dim3(NTH), lds, 0, A, ctr);
    } else {
      hipFuncSetAttribute((const void *)persistent_k<NTH, LT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipLaunchKernelGGL((persistent_k<NTH, LT>), dim3(256 * per_cu), dim3(NTH), lds, 0, A, ctr);
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double bytes = (double)A.B * A.L * 176 * 2 * 5;
  printf("%-44s LT=%3d thr=%4d lds=%6d per_cu=%d: %7.1f us  %5.2f TB/s (%4.1f%% of 8)\n", name, LT, NTH, lds, per_cu, best * 1e3,
         bytes / best / 1e9, bytes / best / 1e9 / 8 * 100);
}

// Full-row variant: a work item = (batch, chunk of LT tokens, ALL channels): every row of every stream is one contiguous run
// (Bt|C 768 B of p's 896-byte row, xc / z / out 352 or 384 B).
template <int NTH, int LT>
__global__ void __launch_bounds__(NTH) fullrow_k(Args A, unsigned *ctr, int persistent) {
  extern __shared__ char smem[];
  __shared__ int item_s;
  constexpr int PPR = 24, TOTAL = LT * PPR, ITERS = (TOTAL + NTH - 1) / NTH;   // 24 16-byte pieces = 384 B per row
  const int rowb = A.xc_pitch < 384 ? 352 : 384;
  while (true) {
    if (threadIdx.x == 0) item_s = (int)atomicAdd(ctr, 1u);
    __syncthreads();
    const int item = item_s;
    if (item >= A.items) break;
    const int chunk = item / A.B, bb = item - chunk * A.B;
    const int64_t tok0 = (int64_t)bb * A.L + (int64_t)chunk * LT;
    uint4 a[ITERS], b[ITERS], c[ITERS], d[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int idx = threadIdx.x + it * NTH, row = idx / PPR, cb = (idx % PPR) * 16;
      const bool ok = idx < TOTAL && cb < rowb;
      const uint4 zz = make_uint4(0, 0, 0, 0);
      a[it] = idx < TOTAL ? *reinterpret_cast<const uint4 *>(A.p + (tok0 + row) * A.p_pitch + cb) : zz;
      b[it] = idx < TOTAL ? *reinterpret_cast<const uint4 *>(A.p + (tok0 + row) * A.p_pitch + A.c_off + cb) : zz;
      c[it] = ok ? *reinterpret_cast<const uint4 *>(A.xc + (tok0 + row) * A.xc_pitch + cb) : zz;
      d[it] = ok ? *reinterpret_cast<const uint4 *>(A.xz + (tok0 + row) * A.xz_pitch + A.z_off + cb) : zz;
    }
    if (smem[threadIdx.x & 15] == 77) a[0].x ^= 1;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int idx = threadIdx.x + it * NTH, row = idx / PPR, cb = (idx % PPR) * 16;
      if (idx < TOTAL && cb < rowb) {
        uint4 o = make_uint4(a[it].x ^ b[it].x ^ c[it].x ^ d[it].x, a[it].y + b[it].y + c[it].y + d[it].y, a[it].z ^ d[it].z,
                             b[it].w ^ c[it].w);
        *reinterpret_cast<uint4 *>(A.out + (tok0 + row) * A.xc_pitch + cb) = o;
      }
    }
    if (!persistent) break;
  }
}

template <int NTH, int LT> void run_full(const char *name, Args A, unsigned *ctr, int lds, int per_cu) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  A.LT = LT; A.nchunks = A.L / LT; A.items = A.nchunks * A.B;
  float best = 1e9;
  hipFuncSetAttribute((const void *)fullrow_k<NTH, LT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int rep = 0; rep < 6; ++rep) {
    flush_caches();
    hipMemsetAsync(ctr, 0, 4, 0);
    hipEventRecord(e0);
    hipLaunchKernelGGL((fullrow_k<NTH, LT>), dim3(per_cu ? 256 * per_cu : A.items), dim3(NTH), lds, 0, A, ctr, per_cu ? 1 : 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double bytes = (double)A.B * A.L * 176 * 2 * 5;
  printf("%-44s LT=%3d thr=%4d lds=%6d per_cu=%d: %7.1f us  %5.2f TB/s (%4.1f%% of 8)\n", name, LT, NTH, lds, per_cu, best * 1e3,
         bytes / best / 1e9, bytes / best / 1e9 / 8 * 100);
}

int main(int argc, char **argv) {
  Args A{};
  A.B = argc > 1 ? atoi(argv[1]) : 32; A.L = 4096; A.ctiles = 3;
  const int64_t T = (int64_t)A.B * A.L;
  char *p, *xc, *xz, *out; unsigned *ctr;
  for (int aligned = 0; aligned < 2; ++aligned) {
    A.p_pitch = 896; A.c_off = 384;
    A.xc_pitch = aligned ? 384 : 352; A.xz_pitch = aligned ? 768 : 704; A.z_off = aligned ? 384 : 352;
    hipMalloc(&p, T * A.p_pitch + 4096); hipMalloc(&xc, T * A.xc_pitch + 4096); hipMalloc(&xz, T * A.xz_pitch + 4096);
    hipMalloc(&out, T * A.xc_pitch + 4096); hipMalloc(&ctr, 64);
    hipMemset(p, 1, T * A.p_pitch); hipMemset(xc, 2, T * A.xc_pitch); hipMemset(xz, 3, T * A.xz_pitch);
    A.p = p; A.xc = xc; A.xz = xz; A.out = out;
    printf("--- B=%d, rows %s ---\n", A.B, aligned ? "padded to 128-byte multiples" : "as in the model (352 / 704 B)");
    run<1024, 128>("one-shot, 2 WG/CU (the kernel's geometry)", 0, A, ctr, 70000, 0);
    run<1024, 128>("one-shot, 1 WG/CU", 0, A, ctr, 100000, 0);
    run<512, 128>("one-shot 512 thr, 4 WG/CU", 0, A, ctr, 36000, 0);
    run<512, 128>("one-shot 512 thr, 2 WG/CU", 0, A, ctr, 70000, 0);
    run<512, 64>("one-shot 512 thr LT 64, 4 WG/CU", 0, A, ctr, 36000, 0);
    run<256, 64>("one-shot 256 thr LT 64, 8 WG/CU", 0, A, ctr, 18000, 0);
    run<1024, 128>("persistent + prefetch, 2 WG/CU", 1, A, ctr, 70000, 2);
    run<1024, 128>("persistent + prefetch, 1 WG/CU", 1, A, ctr, 100000, 1);
    run<512, 128>("persistent + prefetch 512 thr, 4 WG/CU", 1, A, ctr, 36000, 4);
    run<512, 128>("persistent + prefetch 512 thr, 2 WG/CU", 1, A, ctr, 70000, 2);
    run<512, 64>("persistent + prefetch 512 thr LT 64, 4/CU", 1, A, ctr, 36000, 4);
    run_full<768, 64>("FULL ROWS one-shot 768 thr, 2 WG/CU", A, ctr, 70000, 0);
    run_full<768, 64>("FULL ROWS one-shot 768 thr, 1 WG/CU", A, ctr, 100000, 0);
    run_full<768, 32>("FULL ROWS one-shot 768 thr LT 32, 2 WG/CU", A, ctr, 70000, 0);
    run_full<384, 32>("FULL ROWS one-shot 384 thr LT 32, 4 WG/CU", A, ctr, 36000, 0);
    run_full<768, 128>("FULL ROWS one-shot 768 thr LT 128, 1 WG/CU", A, ctr, 100000, 0);
    run_full<768, 64>("FULL ROWS persistent 768 thr, 2 WG/CU", A, ctr, 70000, 2);
    run_full<768, 32>("FULL ROWS persistent 768 thr LT 32, 2 WG/CU", A, ctr, 70000, 2);
    hipFree(p); hipFree(xc); hipFree(xz); hipFree(out); hipFree(ctr);
  }
  return 0;
}
