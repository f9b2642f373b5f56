// Selective scan with the post-scan skip + gate fused in (gfx950), forward and backward, and the single-token decode step.
//
// Reference: SelectiveLinearAttention.forward, /root/reference/src/model/core.py:388-396 -
//   s_t = exp(delta_t*A)*s_{t-1} + Bt_t,  y_t = C_t*s_t                      (:337-353, the recurrence)
//   out_t = (y_t + D*xc_t) * silu(z_t)                                        (:395-396)
// The stand-alone pair (selective_scan.hip + apertis_ssm_gate_*) writes y in fp32, reads it back for the gate and does
// the same with dy in the backward: half of the scan's HBM bytes.  Here y never reaches HBM: the forward writes the gated
// output in the activation dtype, the backward recomputes y from the states it rebuilds anyway.
//
// Algorithmic bytes per token (T = B*L, e = bytes of the activation dtype; SURVEY.md 8(d) "fused epilogue variant"):
//   forward   Dn*e*(Bt + C + xc + z + out) + 4h          = 5*Dn*e + 4h
//   backward  Dn*e*(Bt + C + xc + z + dout) reads + Dn*e*(dBt + dC + dxc + dz) writes + 8h = 9*Dn*e + 8h
//
// Work item = (batch, chunk of 64 tokens, ALL channels): CW groups of 64 channels side by side, so every row of every
// tensor is ONE contiguous run (Bt|C 768 of p's 896-byte rows, xc / z / out 352 B at Dn = 176).  Measured with
// tools/probes/scan_stream.hip (pure loads + stores of this kernel's five streams, B=32, L=4096): 64-channel tiles - 128-byte
// row pieces, the layout of selective_scan.hip - stream at 3.5 TB/s whatever the occupancy, whole rows at 4.8 TB/s.
// lane = channel, wave = (channel group, segment of 64/NS tokens); the Bt / C tiles go HBM -> LDS as whole row segments and
// each thread walks its LDS column; everything elementwise (xc, z, dout in; out, dz, dxc out) is touched row-major in
// 16-byte pieces straight from / to global memory; only y (forward) resp. dv and y (backward) cross from the column walk
// to the row-major side, through an fp32 LDS tile.
//
// Carry between chunks (64 chunks per 4096-token sequence), two forms with the SAME arithmetic and bits:
//   mode 0  two launches: a state pass (mode 2 of the same kernel) writes every chunk's aggregate, the replay composes its
//           carry from them (the first pass reads Bt - resp. C, dout, z - a second time);
//   mode 1  ONE launch: work-groups take their item from a ticket counter in chunk-major order, compute the chunk
//           aggregate from the Bt tile first and PUBLISH it (8-byte {epoch, value} granules, one agent-scope store each:
//           cdna_hip_programming.md Guideline 16, form R2) while their other loads are still in flight; the last chunk
//           of every super-chunk of 8 also publishes the super-chunk's composite, so a chunk gathers at most 7 + 7
//           records: the composites of the earlier super-chunks, then the aggregates of the chunks of its own.
//           A work-group only ever waits for lower tickets, which belong to work-groups that have already started and
//           never wait for a higher one: progress does not depend on residency.  The composition order is fixed, so
//           the result is bit-identical to mode 0 and run-to-run.  Waits are bounded; a timeout sets the error word.
#include "scan_lean.h"

namespace {



#ifndef SCAN_GATE_LT
#define SCAN_GATE_LT 64
#endif
constexpr int LTG = SCAN_GATE_LT;   // tokens per item = apertis_scan_gate_chunk_len(): the granularity of h_in and of the aggregates
constexpr int SUP = 8;     // chunks per super-chunk of the two-level look-back


// A tile of LT rows x ROWB bytes as VB-byte pieces in registers: global -> registers (the loads stay in flight), then
// registers -> LDS.  Thread `tid` owns piece (row0 + q*RSTEP, cb0) for q < ITERS: one row / column pair per thread and a
// constant row step, so the only per-thread state a persistent loop has to keep is (row0, cb0) - with a div / mod per piece
// the compiler hoisted ~50 loop invariants and spilled them (344 bytes of scratch per lane).
template <int VB, int ROWB, int LT, int NTH>
struct TileRegs {
  typedef typename vec_bytes<VB>::type V;
  static constexpr int CPR = ROWB / VB, TOTAL = LT * CPR;
  static_assert(NTH % CPR == 0 && TOTAL % NTH == 0, "a thread's pieces share their column");
  static constexpr int RSTEP = NTH / CPR, ITERS = TOTAL / NTH;
  V r[ITERS];
  __device__ static __forceinline__ int row0(int tid) { return tid / CPR; }
  __device__ static __forceinline__ int cb0(int tid) { return (tid % CPR) * VB; }
  __device__ __forceinline__ void load(const char *g, int64_t rsb, int rows_valid, int bytes_valid, int tid) {
    const int r0 = row0(tid), c0 = cb0(tid);
    const char *p = g + (int64_t)r0 * rsb + c0;
#pragma unroll
    for (int q = 0; q < ITERS; ++q) {
      const bool ok = r0 + q * RSTEP < rows_valid && c0 < bytes_valid;
      r[q] = ok ? nt_load<VB>(p + (int64_t)(q * RSTEP) * rsb) : zero_vec<VB>();
    }
  }
  __device__ __forceinline__ void store(char *lds, int tid) const {
    char *p = lds + row0(tid) * ROWB + cb0(tid);
#pragma unroll
    for (int q = 0; q < ITERS; ++q) *reinterpret_cast<V *>(p + q * RSTEP * ROWB) = r[q];
  }
  // LDS tile -> global rows (the mirror of load + store), same piece ownership
  __device__ static __forceinline__ void tile_out(const char *lds, char *g, int64_t rsb, int rows_valid, int bytes_valid, int tid) {
    const int r0 = row0(tid), c0 = cb0(tid);
    if (c0 >= bytes_valid) return;
    const char *l = lds + r0 * ROWB + c0;
    char *p = g + (int64_t)r0 * rsb + c0;
#pragma unroll
    for (int q = 0; q < ITERS; ++q)
      if (r0 + q * RSTEP < rows_valid) nt_store<VB>(p + (int64_t)(q * RSTEP) * rsb, *reinterpret_cast<const V *>(l + q * RSTEP * ROWB));
  }
};

// A row-major piece of a tile held by one thread: EPC consecutive channels of one token (VB bytes).
template <typename T, int VB> struct Piece {
  typedef typename vec_bytes<VB>::type V;
  static constexpr int EPC = VB / (int)sizeof(T);
  __device__ static __forceinline__ void unpack(const V &v, float (&f)[EPC]) {
    T e[EPC];
    __builtin_memcpy(e, &v, VB);
#pragma unroll
    for (int k = 0; k < EPC; ++k) f[k] = to_f32(e[k]);
  }
  __device__ static __forceinline__ V pack(const float (&f)[EPC]) {
    T e[EPC];
#pragma unroll
    for (int k = 0; k < EPC; ++k) e[k] = from_f32<T>(f[k]);
    V v;
    __builtin_memcpy(&v, e, VB);
    return v;
  }
};

// fp32 y kept in the slots of the Bt / C tiles that produced it.  bf16 tiles: the high half of the fp32 word goes where
// Bt[t][c] was, the low half where C[t][c] was - each thread overwrites only elements it alone has read, so the column
// walk needs neither a barrier nor 16 registers to park y before the row-major epilogue picks it up.  fp32 tiles: y simply
// replaces Bt[t][c].
template <typename T> __device__ __forceinline__ void y_put(T *bt, T *cc, int idx, float v);
template <> __device__ __forceinline__ void y_put<float>(float *bt, float *, int idx, float v) { bt[idx] = v; }
template <> __device__ __forceinline__ void y_put<bf16_t>(bf16_t *bt, bf16_t *cc, int idx, float v) {
  const uint32_t u = __float_as_uint(v);
  reinterpret_cast<uint16_t *>(bt)[idx] = (uint16_t)(u >> 16);
  reinterpret_cast<uint16_t *>(cc)[idx] = (uint16_t)(u & 0xffffu);
}
template <typename T, int EPC> struct YGet;
template <int EPC> struct YGet<float, EPC> {
  __device__ static __forceinline__ void get(const float *bt, const float *, int idx, float (&y)[EPC]) {
#pragma unroll
    for (int k = 0; k < EPC; ++k) y[k] = bt[idx + k];
  }
};
template <int EPC> struct YGet<bf16_t, EPC> {
  __device__ static __forceinline__ void get(const bf16_t *bt, const bf16_t *cc, int idx, float (&y)[EPC]) {
    uint16_t hi[EPC], lo[EPC];
    __builtin_memcpy(hi, reinterpret_cast<const uint16_t *>(bt) + idx, EPC * 2);
    __builtin_memcpy(lo, reinterpret_cast<const uint16_t *>(cc) + idx, EPC * 2);
#pragma unroll
    for (int k = 0; k < EPC; ++k) y[k] = __uint_as_float(((uint32_t)hi[k] << 16) | (uint32_t)lo[k]);
  }
};

// geometry of a work-group: CW channel groups x NS token segments (waves), 64 tokens
template <int CW> struct Geo {
#ifndef SCAN_GATE_NS3
#define SCAN_GATE_NS3 4
#endif
  static constexpr int NS = (CW <= 2 ? 8 : SCAN_GATE_NS3) * LTG / 64, CWC = 64 * CW, NTH = CWC * NS, TS = LTG / NS;
};

// Compose the published records [j0, j1) (granule index = record index) in composition order into (P, S):
// ascending, or descending when `reverse`.  The loads of up to QMAX records are in flight together (a record per round
// trip made a 7-record gather the longest phase of the kernel); spins (bounded) until every granule of the batch
// carries this launch's epoch.
__device__ __forceinline__ void gather_published(const gran_t *gbase, int stride, int j0, int j1, bool reverse, int cl,
                                                 bool chan_ok, uint32_t epoch, int *err, float &P, float &S) {
  constexpr int QMAX = 4;
  P = 1.f; S = 0.f;
  for (int k0 = 0; k0 < j1 - j0; k0 += QMAX) {
    gran_t gp[QMAX], gs[QMAX];
    const int nb = min(QMAX, j1 - j0 - k0);
    unsigned spins = 0;
    while (true) {
      bool ok = true;
#pragma unroll
      for (int u = 0; u < QMAX; ++u)
        if (u < nb && chan_ok) {
          const int j = reverse ? (j1 - 1 - k0 - u) : (j0 + k0 + u);
          gp[u] = gran_load(gbase + ((int64_t)j * stride + cl) * 2 + 0);
          gs[u] = gran_load(gbase + ((int64_t)j * stride + cl) * 2 + 1);
        }
#pragma unroll
      for (int u = 0; u < QMAX; ++u)
        if (u < nb && chan_ok) ok = ok && (uint32_t)(gp[u] >> 32) == epoch && (uint32_t)(gs[u] >> 32) == epoch;
      if (__all(ok)) break;
      if (++spins > (1u << 18)) { if ((cl & 63) == 0) atomicOr(err, 1); break; }
      __builtin_amdgcn_s_sleep(2);
    }
#pragma unroll
    for (int u = 0; u < QMAX; ++u)
      if (u < nb && chan_ok) {
        const float pj = __uint_as_float((uint32_t)gp[u]), sj = __uint_as_float((uint32_t)gs[u]);
        S = fmaf(pj, S, sj);
        P *= pj;
      }
  }
}
// the same composition from the aggregate array of the two-launch form (loads batched the same way)
__device__ __forceinline__ void fold_agg(const float2 *agg, int64_t base, int64_t Dn, int j0, int j1, bool reverse, bool chan_ok,
                                         float &P, float &S) {
  constexpr int QMAX = 8;
  P = 1.f; S = 0.f;
  if (!chan_ok) return;
  for (int k0 = 0; k0 < j1 - j0; k0 += QMAX) {
    float2 r[QMAX];
    const int nb = min(QMAX, j1 - j0 - k0);
#pragma unroll
    for (int u = 0; u < QMAX; ++u)
      if (u < nb) r[u] = agg[base + (int64_t)(reverse ? (j1 - 1 - k0 - u) : (j0 + k0 + u)) * Dn];
#pragma unroll
    for (int u = 0; u < QMAX; ++u)
      if (u < nb) { S = fmaf(r[u].x, S, r[u].y); P *= r[u].x; }
  }
}

// item of this work-group: (chunk, batch, channel super-tile) from the grid (modes 0 / 2) or from the ticket counter in
// chunk-major order (mode 1; `reverse`: chunks right to left)
template <int MODE>
__device__ __forceinline__ void take_item(GateWsHead *head, uint32_t epoch, int *item_s, const ScanDims &d, int ncs, bool reverse,
                                          int tid, int &chunk, int &cs, int &b) {
  if constexpr (MODE == 1) {
    if (tid == 0) {
      const unsigned t = atomicAdd(&head->ctr[epoch & 1], 1u);
      if (t == 0) atomicExch(&head->ctr[(epoch + 1) & 1], 0u);   // the next launch's counter (its last user has finished)
      *item_s = (int)t;
    }
    __syncthreads();
    const int item = *item_s, bcs = (int)d.B * ncs;
    const int k = item / bcs, r = item - k * bcs;
    chunk = reverse ? d.nchunks - 1 - k : k;
    b = r / ncs;
    cs = r - b * ncs;
  } else {
    chunk = blockIdx.x; cs = blockIdx.y; b = blockIdx.z;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// (A persistent form of the forward - a work-group walks items and issues its next item's loads one phase ahead - was
// built and measured with tools/probes/scan_gate_probe.hip at B=32, L=4096, Dn=176: the wait for the first tile goes from
// 5.6 to 0.9 us per item, but the loads parked in registers across the loop either spill (two work-groups per CU at 80
// VGPRs: 134 us) or leave one work-group per CU whose column walks then take twice as long (91 us), against 73 us for one
// item per work-group with warm caches.  Removed.
//  Two more forms measured later in round 2 (same probe, cold caches, B=32; this kernel: 84.0 us, 11.5 us per item, 321
//  work-groups in flight): (i) work-groups that keep taking tickets (the next one drawn late in the item, nothing
//  prefetched, the thread index laundered per trip so nothing per-thread is hoisted): 365 in flight but 14.7 us per item -
//  every phase, the pure-compute replay included, runs slower with two work-groups truly co-resident all the time - 90.2 us;
//  (ii) the same with ONE walk per item (y from a zero state to the tile, C*prod(a) kept in registers, y += q*H after the
//  look-back): the read-modify-write of the split 2-byte y slots costs 1.9 us and the late xc / z loads 0.8 us - 87.6 us.
//  The kernel is bound by per-CU instruction issue (walks, silu, bf16 packing) and latency together, not by slots.)
struct ItemPos { int chunk, cs, b, id; };
template <int MODE>
__device__ __forceinline__ ItemPos decode_item(int item, const ScanDims &d, int ncs, bool reverse) {
  const int bcs = (int)d.B * ncs;
  const int k = item / bcs, r = item - k * bcs;
  ItemPos p;
  p.id = item;
  p.chunk = reverse ? d.nchunks - 1 - k : k;
  p.b = r / ncs;
  p.cs = r - p.b * ncs;
  return p;
}

// forward: out = (C*s + D*xc) * silu(z).   MODE 0: replay, carry from agg[]; 1: single launch; 2: state pass (writes agg[])
//   LDS: Bt and C tiles [64][CWC] (staged as whole rows), later overwritten by the fp32 y tile; delta tile; a [NS][CWC]
//   table of segment aggregates; two [CWC] look-back records.
template <typename T, int VB, int CW, int MODE, int MINW>
__global__ void __launch_bounds__(Geo<CW>::NTH, MINW)
scan_gate_fwd_k(const float *__restrict__ dlt, const float *__restrict__ A_log, const T *__restrict__ Bt, int64_t bt_rs,
                const T *__restrict__ C, int64_t c_rs, const T *__restrict__ xc, int64_t xc_rs, const T *__restrict__ z,
                int64_t z_rs, const float *__restrict__ Dv, const float *__restrict__ h0, float2 *__restrict__ agg,
                GateWsHead *__restrict__ head, gran_t *__restrict__ gran, uint32_t epoch, float *__restrict__ h_in,
                float *__restrict__ h_last, T *__restrict__ out, int64_t out_rs, ScanDims d, int ncs, int total) {
  typedef Geo<CW> G;
  constexpr int NS = G::NS, CWC = G::CWC, NTH = G::NTH, TS = G::TS, LT = LTG;
  constexpr int ROWB = CWC * sizeof(T);
  typedef TileRegs<VB, ROWB, LT, NTH> TR;
  typedef Piece<T, VB> PC;
  constexpr int EPC = PC::EPC;
  const int HTC = CW * d.HT;                                        // heads covered by the tile
  const int HS = ncs == 1 ? (int)d.h : HTC;                         // row pitch of the delta tile in LDS
  constexpr int DLI = (LT * 16 + NTH - 1) / NTH;                    // delta values per thread (at most 16 heads per 64 channels ... N >= 4)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T *bt = reinterpret_cast<T *>(smem);
  T *cc = bt + LT * CWC;
  float *dl = reinterpret_cast<float *>(smem + 2 * LT * ROWB);
  float2 *segs = reinterpret_cast<float2 *>(dl + LT * HTC);
  float2 *lkA = segs + NS * CWC, *lkX = lkA + CWC;
  float *dtab = reinterpret_cast<float *>(lkX + CWC);
  int *item_s = reinterpret_cast<int *>(dtab + CWC);               // [0]: ticket; [2]: probe id

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, cw = w / NS, seg = w - cw * NS, cl = cw * 64 + lane;
  const int hh = cl >> d.log2N;
  int cur_item;
  if constexpr (MODE == 1) {
    if (tid == 0) {
      const unsigned t = atomicAdd(&head->ctr[epoch & 1], 1u);
      if (t == 0) atomicExch(&head->ctr[(epoch + 1) & 1], 0u);   // the next launch's counter (its last user has finished)
      item_s[0] = (int)t;
    }
    __syncthreads();
    cur_item = item_s[0];
  } else {
    cur_item = blockIdx.x;
  }
  if (cur_item >= total) return;
  const ItemPos it = decode_item<MODE>(cur_item, d, ncs, false);

  TR rb, rc, rx, rz;
  float dreg[DLI];
  // loads of an item's tiles (issued one item ahead)
  auto geom = [&](const ItemPos &p, int &c0, int64_t &tok0, int &rows_valid, int &vb) {
    c0 = p.cs * CWC;
    const int64_t t0 = (int64_t)p.chunk * LT;
    rows_valid = (int)min((int64_t)LT, d.L - t0);
    tok0 = (int64_t)p.b * d.L + t0;
    vb = (int)min((int64_t)CWC, d.Dn - c0) * (int)sizeof(T);
  };
  auto load_bt = [&](const ItemPos &p) {
    int c0, rows_valid, vb; int64_t tok0;
    geom(p, c0, tok0, rows_valid, vb);
    rb.load(reinterpret_cast<const char *>(Bt + tok0 * bt_rs + c0), bt_rs * sizeof(T), rows_valid, vb, tid);
    if (ncs == 1) {                                                 // all heads: the tile's delta values are one contiguous run
      const int nv = rows_valid * (int)d.h;
#pragma unroll
      for (int k = 0; k < DLI; ++k) dreg[k] = tid + k * NTH < nv ? dlt[tok0 * d.h + tid + k * NTH] : -INFINITY;
    } else {
      const int head0 = c0 >> d.log2N;
#pragma unroll
      for (int k = 0; k < DLI; ++k) {
        const int idx = tid + k * NTH, t = idx / HTC, hx = idx - t * HTC;
        dreg[k] = (idx < LT * HTC && t < rows_valid && head0 + hx < d.h) ? dlt[(tok0 + t) * d.h + head0 + hx] : -INFINITY;
      }
    }
  };
  auto load_c = [&](const ItemPos &p) {
    int c0, rows_valid, vb; int64_t tok0;
    geom(p, c0, tok0, rows_valid, vb);
    rc.load(reinterpret_cast<const char *>(C + tok0 * c_rs + c0), c_rs * sizeof(T), rows_valid, vb, tid);
  };
  auto load_xz = [&](const ItemPos &p) {
    int c0, rows_valid, vb; int64_t tok0;
    geom(p, c0, tok0, rows_valid, vb);
    rx.load(reinterpret_cast<const char *>(xc + tok0 * xc_rs + c0), xc_rs * sizeof(T), rows_valid, vb, tid);
    rz.load(reinterpret_cast<const char *>(z + tok0 * z_rs + c0), z_rs * sizeof(T), rows_valid, vb, tid);
  };
  load_bt(it);
  if constexpr (MODE != 2) { load_c(it); load_xz(it); }

  {
    const int chunk = it.chunk, cs = it.cs, b = it.b;
    const int c0 = cs * CWC, c = c0 + cl;
    const int64_t t0 = (int64_t)chunk * LT;
    const int rows_valid = (int)min((int64_t)LT, d.L - t0);
    const int ch_valid = (int)min((int64_t)CWC, d.Dn - c0);
    const int64_t tok0 = (int64_t)b * d.L + t0;
    const bool chan_ok = c < d.Dn;
    const int vb = ch_valid * (int)sizeof(T);
    const float A2 = chan_ok ? -expf(A_log[c]) * LOG2E_F : 0.f;
    if constexpr (MODE != 2) {
      if (tid < CWC) dtab[tid] = (c0 + tid < d.Dn) ? Dv[c0 + tid] : 0.f;
    }
    rb.store(reinterpret_cast<char *>(bt), tid);
#pragma unroll
    for (int k = 0; k < DLI; ++k) {
      const int idx = tid + k * NTH;
      if (idx < LT * HTC) dl[idx] = dreg[k] == -INFINITY ? 0.f : (d.softplus ? softplus_f(dreg[k]) : dreg[k]);
    }
    __syncthreads();   // Bt and delta tiles are in LDS

    float a[TS];
    float P = 1.f, S = 0.f;
#pragma unroll
    for (int i = 0; i < TS; ++i) {
      const int t = seg * TS + i;
      a[i] = __builtin_amdgcn_exp2f(dl[t * HS + hh] * A2);
      S = fmaf(a[i], S, to_f32(bt[t * CWC + cl]));
      P *= a[i];
    }
    segs[seg * CWC + cl] = make_float2(P, S);
    __syncthreads();

    // composite of the segments in front of this wave's
    float Ppre = 1.f, Spre = 0.f;
    for (int s = 0; s < seg; ++s) { const float2 r = segs[s * CWC + cl]; Spre = fmaf(r.x, Spre, r.y); Ppre *= r.x; }
    const int64_t abase = (int64_t)b * d.nchunks * d.Dn + c;
    if constexpr (MODE == 2) {
      if (seg == NS - 1 && chan_ok) agg[abase + (int64_t)chunk * d.Dn] = make_float2(Ppre * P, fmaf(P, Spre, S));
    } else {
      // carry entering the chunk = [composites of the earlier super-chunks] then [aggregates of the earlier chunks of this one]
      const int sup = chunk / SUP, base0 = sup * SUP;
      if constexpr (MODE == 1) {
        const int nrec = d.nchunks + (d.nchunks + SUP - 1) / SUP;
        gran_t *gb = gran + ((int64_t)(b * ncs + cs) * nrec) * CWC * 2;
        if (seg == NS - 1 && chunk + 1 < d.nchunks && chan_ok) {     // publish this chunk's aggregate (the last chunk has no reader)
          gran_store(gb + ((int64_t)chunk * CWC + cl) * 2 + 0, epoch, Ppre * P);
          gran_store(gb + ((int64_t)chunk * CWC + cl) * 2 + 1, epoch, fmaf(P, Spre, S));
        }
        if (seg == 0) {
          float PA, SA;
          gather_published(gb, CWC, base0, chunk, false, cl, chan_ok, epoch, &head->err, PA, SA);
          if (chunk - base0 == SUP - 1 && chunk + 1 < d.nchunks && chan_ok) {   // last chunk of its super-chunk: publish the composite
            float Po = 1.f, So = 0.f;
            for (int s = 0; s < NS; ++s) { const float2 r = segs[s * CWC + cl]; So = fmaf(r.x, So, r.y); Po *= r.x; }
            gran_store(gb + ((int64_t)(d.nchunks + sup) * CWC + cl) * 2 + 0, epoch, PA * Po);
            gran_store(gb + ((int64_t)(d.nchunks + sup) * CWC + cl) * 2 + 1, epoch, fmaf(Po, SA, So));
          }
          lkA[cl] = make_float2(PA, SA);
        } else if (seg == 1) {
          float PX, SX;
          gather_published(gb, CWC, d.nchunks, d.nchunks + sup, false, cl, chan_ok, epoch, &head->err, PX, SX);
          lkX[cl] = make_float2(PX, SX);
        }
      } else {
        if (seg == 0) {
          float PA, SA;
          fold_agg(agg, abase, d.Dn, base0, chunk, false, chan_ok, PA, SA);
          lkA[cl] = make_float2(PA, SA);
        } else if (seg == 1) {
          float PX = 1.f, SX = 0.f;
          for (int i = 0; i < sup; ++i) {
            float Pi, Si;
            fold_agg(agg, abase, d.Dn, i * SUP, i * SUP + SUP, false, chan_ok, Pi, Si);
            SX = fmaf(Pi, SX, Si);
            PX *= Pi;
          }
          lkX[cl] = make_float2(PX, SX);
        }
      }
      rc.store(reinterpret_cast<char *>(cc), tid);
      __syncthreads();

      float hcar = (h0 && chan_ok) ? h0[(int64_t)b * d.Dn + c] : 0.f;
      { const float2 rX = lkX[cl], rA = lkA[cl]; hcar = fmaf(rX.x, hcar, rX.y); hcar = fmaf(rA.x, hcar, rA.y); }
      if (seg == 0 && chan_ok) h_in[abase + (int64_t)chunk * d.Dn] = hcar;    // state entering the chunk, saved for the backward
      hcar = fmaf(Ppre, hcar, Spre);
      float hst = hcar;
#pragma unroll
      for (int i = 0; i < TS; ++i) {
        const int t = seg * TS + i;
        hst = fmaf(a[i], hst, to_f32(bt[t * CWC + cl]));
        y_put<T>(bt, cc, t * CWC + cl, to_f32(cc[t * CWC + cl]) * hst);
      }
      if (h_last && chunk == d.nchunks - 1 && seg == NS - 1 && chan_ok) h_last[(int64_t)b * d.Dn + c] = hst;
      __syncthreads();   // y (in the Bt / C slots) is complete

      // row-major epilogue: out = (y + D*xc) * silu(z), 16-byte pieces straight to global memory
      char *og = reinterpret_cast<char *>(out + tok0 * out_rs + c0);
      const int prow0 = TR::row0(tid), cb = TR::cb0(tid);
#pragma unroll
      for (int q = 0; q < TR::ITERS; ++q) {
        const int row = prow0 + q * TR::RSTEP;
        if (row < rows_valid && cb < vb) {
          const int e0 = cb / (int)sizeof(T);
          float xv[EPC], zv[EPC], yk[EPC], o[EPC];
          PC::unpack(rx.r[q], xv);
          PC::unpack(rz.r[q], zv);
          YGet<T, EPC>::get(bt, cc, row * CWC + e0, yk);
#pragma unroll
          for (int k = 0; k < EPC; ++k) {
            const float dx = dtab[e0 + k] * xv[k];
            const float v = yk[k] + dx;
            o[k] = v * silu_g(zv[k]);
          }
          nt_store<VB>(og + (int64_t)row * out_rs * sizeof(T) + cb, PC::pack(o));
        }
      }
    }
  }
}

// N consecutive fp32 of an LDS tile (N a power of two, the address aligned to 4*N bytes) as 16- / 8- / 4-byte accesses
template <int N> struct lds_vec {
  static_assert(N == 1 || N == 2 || N % 4 == 0, "piece widths are powers of two");
  __device__ static __forceinline__ void load(const float *p, float (&v)[N]) {
    if constexpr (N % 4 == 0) {
#pragma unroll
      for (int k = 0; k < N / 4; ++k) {
        const float4 t = reinterpret_cast<const float4 *>(p)[k];
        v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
      }
    } else if constexpr (N == 2) {
      const float2 t = *reinterpret_cast<const float2 *>(p);
      v[0] = t.x; v[1] = t.y;
    } else {
      v[0] = p[0];
    }
  }
  __device__ static __forceinline__ void store(float *p, const float (&v)[N]) {
    if constexpr (N % 4 == 0) {
#pragma unroll
      for (int k = 0; k < N / 4; ++k) reinterpret_cast<float4 *>(p)[k] = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
    } else if constexpr (N == 2) {
      *reinterpret_cast<float2 *>(p) = make_float2(v[0], v[1]);
    } else {
      p[0] = v[0];
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------
// backward.  dv = dout*silu(z) (= d/dy) is formed row-major from the 16-byte pieces of dout and z as they arrive and
// lives in an fp32 LDS tile; the column walk turns that tile into y in place; dz = dout*silu'(z)*(y + D*xc) and
// dxc = dv*D leave row-major straight from registers; dBt / dC go through the Bt / C tiles as whole rows.
//   u_t = dv_t*C_t,  mu_t = a_t*(u_t + mu_{t+1});  reverse aggregates (P = prod a, M = mu at chunk start from zero).
// MODE 0: replay, carry from agg[]; 1: single launch (chunks right to left); 2: state pass (writes agg[]).
template <typename T, int VB, int CW, int MODE>
__global__ void __launch_bounds__(Geo<CW>::NTH)
scan_gate_bwd_k(const float *__restrict__ dlt, const float *__restrict__ A_log, const T *__restrict__ Bt, int64_t bt_rs,
                const T *__restrict__ C, int64_t c_rs, const T *__restrict__ xc, int64_t xc_rs, const T *__restrict__ z,
                int64_t z_rs, const float *__restrict__ Dv, const T *__restrict__ dout, int64_t do_rs,
                const float *__restrict__ h_in, float2 *__restrict__ agg, GateWsHead *__restrict__ head,
                gran_t *__restrict__ gran, uint32_t epoch, T *__restrict__ dBt, int64_t dbt_rs, T *__restrict__ dC,
                int64_t dc_rs, int64_t store_w, T *__restrict__ dxc, int64_t dxc_rs, T *__restrict__ dz, int64_t dz_rs,
                float *__restrict__ d_dlt, float *__restrict__ part, ScanDims d, int ncs) {
  typedef Geo<CW> G;
  constexpr int NS = G::NS, CWC = G::CWC, NTH = G::NTH, TS = G::TS, LT = LTG;
  constexpr int ROWB = CWC * sizeof(T);
  typedef TileRegs<VB, ROWB, LT, NTH> TR;
  typedef Piece<T, VB> PC;
  constexpr int EPC = PC::EPC;
  const int HTC = CW * d.HT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T *bt = reinterpret_cast<T *>(smem);
  T *cc = bt + LT * CWC;
  float *pt = reinterpret_cast<float *>(smem);                      // dv*xc tile [64][CWC] fp32, over bt + cc once they have left
  float *dvt = reinterpret_cast<float *>(smem + 2 * LT * ROWB);    // dv, then y: [64][CWC] fp32
  float *dl = dvt + LT * CWC;
  float *ddl = dl + LT * HTC;
  float *segs = ddl + LT * HTC;                                     // [NS][CWC][3] = (P, S, M) of the token segments
  float2 *lkA = reinterpret_cast<float2 *>(segs + NS * CWC * 3), *lkX = lkA + CWC;
  float2 *red = lkX + CWC;                                          // [NS][CWC] partial sums (dA_log, dD)
  float *dtab = reinterpret_cast<float *>(red + NS * CWC);
  int *item_s = reinterpret_cast<int *>(dtab + CWC);   // [0] ticket, [1] probe id

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, cw = w / NS, seg = w - cw * NS, cl = cw * 64 + lane;
  int chunk, cs, b;
  take_item<MODE>(head, epoch, item_s, d, ncs, true, tid, chunk, cs, b);
  const int c0 = cs * CWC, c = c0 + cl;
  const int64_t t0 = (int64_t)chunk * LT;
  const int rows_valid = (int)min((int64_t)LT, d.L - t0);
  const int ch_valid = (int)min((int64_t)CWC, d.Dn - c0);
  const int ch_store = (int)max((int64_t)0, min((int64_t)CWC, store_w - c0));   // dBt / dC zero-extended to the padded slice width
  const int64_t tok0 = (int64_t)b * d.L + t0;
  const int head0 = c0 >> d.log2N;
  const bool chan_ok = c < d.Dn;
  const int vb = ch_valid * (int)sizeof(T);

  TR rc, rg, rz, rb, rx;
  rc.load(reinterpret_cast<const char *>(C + tok0 * c_rs + c0), c_rs * sizeof(T), rows_valid, vb, tid);
  rg.load(reinterpret_cast<const char *>(dout + tok0 * do_rs + c0), do_rs * sizeof(T), rows_valid, vb, tid);
  rz.load(reinterpret_cast<const char *>(z + tok0 * z_rs + c0), z_rs * sizeof(T), rows_valid, vb, tid);
  stage_delta<LT, NTH>(dl, dlt, tok0, rows_valid, head0, (int)d.h, HTC, d.softplus, tid);
  const float Ac = chan_ok ? -expf(A_log[c]) : 0.f;
  const float A2 = Ac * LOG2E_F;
  const int64_t abase = (int64_t)b * d.nchunks * d.Dn + c;
  float hcar = 0.f;
  if constexpr (MODE != 2) {
    rb.load(reinterpret_cast<const char *>(Bt + tok0 * bt_rs + c0), bt_rs * sizeof(T), rows_valid, vb, tid);
    rx.load(reinterpret_cast<const char *>(xc + tok0 * xc_rs + c0), xc_rs * sizeof(T), rows_valid, vb, tid);
    if (tid < CWC) dtab[tid] = (c0 + tid < d.Dn) ? Dv[c0 + tid] : 0.f;
    hcar = chan_ok ? h_in[abase + (int64_t)chunk * d.Dn] : 0.f;
  }
  rc.store(reinterpret_cast<char *>(cc), tid);
#pragma unroll
  for (int it = 0; it < TR::ITERS; ++it) {                          // dv tile, row-major
    {
      const int row = TR::row0(tid) + it * TR::RSTEP, e0 = TR::cb0(tid) / (int)sizeof(T);
      float gv[EPC], zv[EPC];
      PC::unpack(rg.r[it], gv);
      PC::unpack(rz.r[it], zv);
#pragma unroll
      for (int k = 0; k < EPC; ++k) dvt[row * CWC + e0 + k] = gv[k] * silu_g(zv[k]);
    }
    __builtin_amdgcn_sched_barrier(0);                              // one piece at a time: interleaved, their temporaries spill
  }
  __syncthreads();   // C, dv, delta tiles are in LDS

  // a_t = exp2(delta_t*A2) is recomputed wherever it is needed (three times per token) instead of living in 16 registers:
  // the kernel runs one work-group per CU and every array that spills turns its column walk into scratch traffic
  // (measured: 9.1 us for the adjoint loop with 208 bytes of scratch per lane)
  const int hh = cl >> d.log2N;
  auto a_of = [&](int t) { return __builtin_amdgcn_exp2f(dl[t * HTC + hh] * A2); };
  float P = 1.f, M = 0.f;
#pragma unroll 4
  for (int i = TS - 1; i >= 0; --i) {
    const int t = seg * TS + i;
    const float av = a_of(t);
    const float u = dvt[t * CWC + cl] * to_f32(cc[t * CWC + cl]);
    M = av * (u + M);
    P *= av;
  }
  segs[(seg * CWC + cl) * 3 + 0] = P;
  segs[(seg * CWC + cl) * 3 + 2] = M;
  __syncthreads();

  // composite of the segments behind this wave's (right to left)
  float Psuf = 1.f, Msuf = 0.f;
  for (int s = NS - 1; s > seg; --s) {
    const float px = segs[(s * CWC + cl) * 3 + 0], mx = segs[(s * CWC + cl) * 3 + 2];
    Msuf = fmaf(px, Msuf, mx);
    Psuf *= px;
  }
  if constexpr (MODE == 2) {
    if (seg == 0 && chan_ok) agg[abase + (int64_t)chunk * d.Dn] = make_float2(Psuf * P, fmaf(P, Msuf, M));
    return;
  }
  // mu entering from the right = [composites of the later super-chunks] then [aggregates of the later chunks of this one]
  const int nsup = (d.nchunks + SUP - 1) / SUP, sup = chunk / SUP, top = min(sup * SUP + SUP, d.nchunks);
  if constexpr (MODE == 1) {
    const int nrec = d.nchunks + nsup;
    gran_t *gb = gran + ((int64_t)(b * ncs + cs) * nrec) * CWC * 2;
    if (seg == 0 && chunk > 0 && chan_ok) {                        // the whole chunk: this wave's segment after everything behind it
      gran_store(gb + ((int64_t)chunk * CWC + cl) * 2 + 0, epoch, Psuf * P);
      gran_store(gb + ((int64_t)chunk * CWC + cl) * 2 + 1, epoch, fmaf(P, Msuf, M));
    }
    if (seg == 1) {
      float PA, MA;
      gather_published(gb, CWC, chunk + 1, top, true, cl, chan_ok, epoch, &head->err, PA, MA);
      if (chunk == sup * SUP && sup > 0 && chan_ok) {              // first chunk of its super-chunk: publish the composite
        float Po = 1.f, Mo = 0.f;
        for (int s = NS - 1; s >= 0; --s) {
          const float px = segs[(s * CWC + cl) * 3 + 0], mx = segs[(s * CWC + cl) * 3 + 2];
          Mo = fmaf(px, Mo, mx);
          Po *= px;
        }
        gran_store(gb + ((int64_t)(d.nchunks + sup) * CWC + cl) * 2 + 0, epoch, PA * Po);
        gran_store(gb + ((int64_t)(d.nchunks + sup) * CWC + cl) * 2 + 1, epoch, fmaf(Po, MA, Mo));
      }
      lkA[cl] = make_float2(PA, MA);
    } else if (seg == 2) {
      float PX, MX;
      gather_published(gb, CWC, d.nchunks + sup + 1, d.nchunks + nsup, true, cl, chan_ok, epoch, &head->err, PX, MX);
      lkX[cl] = make_float2(PX, MX);
    }
  } else {
    if (seg == 1) {
      float PA, MA;
      fold_agg(agg, abase, d.Dn, chunk + 1, top, true, chan_ok, PA, MA);
      lkA[cl] = make_float2(PA, MA);
    } else if (seg == 2) {
      float PX = 1.f, MX = 0.f;
      for (int i = nsup - 1; i > sup; --i) {
        float Pi, Mi;
        fold_agg(agg, abase, d.Dn, i * SUP, min(i * SUP + SUP, d.nchunks), true, chan_ok, Pi, Mi);
        MX = fmaf(Pi, MX, Mi);
        PX *= Pi;
      }
      lkX[cl] = make_float2(PX, MX);
    }
  }
  rb.store(reinterpret_cast<char *>(bt), tid);
  __syncthreads();   // Bt tile and the look-back records are in LDS

  // forward segment aggregates (need Bt) for the states inside the chunk
  float S = 0.f;
#pragma unroll 4
  for (int i = 0; i < TS; ++i) S = fmaf(a_of(seg * TS + i), S, to_f32(bt[(seg * TS + i) * CWC + cl]));
  segs[(seg * CWC + cl) * 3 + 1] = S;
  float mcar = 0.f;
  { const float2 rX = lkX[cl], rA = lkA[cl]; mcar = fmaf(rX.x, mcar, rX.y); mcar = fmaf(rA.x, mcar, rA.y); }
  mcar = fmaf(Psuf, mcar, Msuf);
  __syncthreads();
  for (int s = 0; s < seg; ++s) hcar = fmaf(segs[(s * CWC + cl) * 3 + 0], hcar, segs[(s * CWC + cl) * 3 + 1]);

  // adjoint, right to left:  lambda_t = dv_t*C_t + mu_{t+1};  mu_t = a_t*lambda_t.  The states it needs are rebuilt in two
  // halves of the segment (the second half first, from the state at the midpoint), so only TS/2 of them are live.
  constexpr int HF = TS / 2;
  float hmid = hcar;
#pragma unroll 4
  for (int i = 0; i < HF; ++i) hmid = fmaf(a_of(seg * TS + i), hmid, to_f32(bt[(seg * TS + i) * CWC + cl]));
  float mu = mcar, dA_acc = 0.f;
#pragma unroll 1
  for (int half = 1; half >= 0; --half) {
    const float h0v = half ? hmid : hcar;                  // state entering this half
    float hs[HF];
    float hst = h0v;
#pragma unroll
    for (int i = 0; i < HF; ++i) {
      const int t = seg * TS + half * HF + i;
      hst = fmaf(a_of(t), hst, to_f32(bt[t * CWC + cl]));
      hs[i] = hst;
    }
#pragma unroll
    for (int i = HF - 1; i >= 0; --i) {
      const int t = seg * TS + half * HF + i;
      const float av = a_of(t);
      const float dv = dvt[t * CWC + cl], Cv = to_f32(cc[t * CWC + cl]);
      const float lam = fmaf(dv, Cv, mu);
      const float hprev = i > 0 ? hs[i - 1] : h0v;
      const float q = lam * hprev * av * Ac;                // da_t * a_t * A
      const float dlv = dl[t * HTC + hh];
      dA_acc = fmaf(q, dlv, dA_acc);
      const float qs = group_sum(q, (int)d.N);
      if ((lane & ((int)d.N - 1)) == 0) ddl[t * HTC + hh] = qs;
      cc[t * CWC + cl] = from_f32<T>(dv * hs[i]);           // dC_t (in place: own column only)
      bt[t * CWC + cl] = from_f32<T>(lam);                  // dBt_t
      dvt[t * CWC + cl] = Cv * hs[i];                       // y_t takes dv_t's place (the epilogue recomputes dv)
      mu = av * lam;
    }
  }
  __syncthreads();   // dBt / dC / y / ddl tiles complete
  TR::tile_out(reinterpret_cast<const char *>(bt), reinterpret_cast<char *>(dBt + tok0 * dbt_rs + c0), dbt_rs * sizeof(T), rows_valid,
               ch_store * (int)sizeof(T), tid);
  TR::tile_out(reinterpret_cast<const char *>(cc), reinterpret_cast<char *>(dC + tok0 * dc_rs + c0), dc_rs * sizeof(T), rows_valid,
               ch_store * (int)sizeof(T), tid);
  for (int idx = tid; idx < LT * HTC; idx += NTH) {
    const int t = idx / HTC, hx = idx - t * HTC;
    if (t < rows_valid && head0 + hx < d.h) {
      float v = ddl[idx];
      if (d.softplus) v *= 1.f - expf(-dl[idx]);            // sigmoid(x) = 1 - exp(-softplus(x))
      d_dlt[(tok0 + t) * d.h + head0 + hx] = v;
    }
  }
  __syncthreads();   // the Bt / C tiles have left: the dv*xc tile takes their place

  // row-major epilogue: dz = dout*silu'(z)*(y + D*xc), dxc = dv*D straight to global memory; dv*xc into the tile
  char *zg = reinterpret_cast<char *>(dz + tok0 * dz_rs + c0), *xg = reinterpret_cast<char *>(dxc + tok0 * dxc_rs + c0);
#pragma unroll
  for (int it = 0; it < TR::ITERS; ++it) {
    {
      const int row = TR::row0(tid) + it * TR::RSTEP, cb = TR::cb0(tid), e0 = cb / (int)sizeof(T);
      float gv[EPC], zv[EPC], xv[EPC], oz[EPC], ox[EPC], yv[EPC], dc[EPC], pv[EPC];
      PC::unpack(rg.r[it], gv);
      PC::unpack(rz.r[it], zv);
      PC::unpack(rx.r[it], xv);
      // the thread's EPC consecutive fp32 of the y tile, the D table and the dv*xc tile move as whole vectors: one element
      // at a time the lanes' 4*EPC-byte stride is an EPC-way bank conflict on every access (8-way at 16-byte bf16 pieces)
      lds_vec<EPC>::load(dvt + row * CWC + e0, yv);
      lds_vec<EPC>::load(dtab + e0, dc);
#pragma unroll
      for (int k = 0; k < EPC; ++k) {
        float f, df;
        silu_both(zv[k], f, df);
        const float dv = gv[k] * f, dcol = dc[k];
        const float dx = dcol * xv[k];
        const float v = yv[k] + dx;
        oz[k] = gv[k] * df * v;
        ox[k] = dv * dcol;
        pv[k] = dv * xv[k];                                 // zero for rows / channels outside the tensor (staged zeros)
      }
      lds_vec<EPC>::store(pt + row * CWC + e0, pv);
      if (row < rows_valid && cb < vb) {
        nt_store<VB>(zg + (int64_t)row * dz_rs * sizeof(T) + cb, PC::pack(oz));
        nt_store<VB>(xg + (int64_t)row * dxc_rs * sizeof(T) + cb, PC::pack(ox));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < TS; ++i) s += pt[(seg * TS + i) * CWC + cl];
    red[seg * CWC + cl] = make_float2(dA_acc, s);
  }
  __syncthreads();
  if (seg < 2 && chan_ok) {                                  // segment-0 waves: dA_log partial, segment-1 waves: dD partial
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NS; ++k) s += seg == 0 ? red[k * CWC + cl].x : red[k * CWC + cl].y;
    part[(((int64_t)b * d.nchunks + chunk) * 2 + seg) * d.Dn + c] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Single-token decode step (reference core.py:364-400 with L = 1 and a cache; generate() core.py:1578-1603):
//   window = [conv_state (k-1 tokens) | xp_new]; the reference keeps the FIRST conv output of that window
//   (core.py:369-373 slices [:, :, :L] of the padded conv), i.e. taps over [0, 0, 0, conv_state[0]] ... - reproduced
//   as is (front-slice quirk); new conv_state = last k-1 tokens of the window;
//   xc = silu(conv + bias)  ->  (caller: x_param_proj, dt)  ->  s = exp(softplus(dt)*A)*s + Bt;  out = (C*s + D*xc)*silu(z)
// Two kernels around the x_param_proj GEMM the caller runs: decode_conv_k (B x Dn threads) and decode_state_k.
template <typename T>
__global__ void __launch_bounds__(256)
decode_conv_k(const T *__restrict__ xp, int64_t xp_rs, const T *conv_state, T *conv_state_out,
              const float *__restrict__ w, const float *__restrict__ bias, T *__restrict__ xc, int64_t B, int64_t Dn, int k) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B * Dn) return;
  const int64_t b = i / Dn, c = i - b * Dn;
  // window of k tokens: conv_state[b, c, 0..k-2] then the new xp; conv1d(padding = k-1) output 0 sees inputs
  // [-(k-1) .. 0] = k-1 zeros and window[0], weighted by taps w[c, 0..k-1]: only the LAST tap meets a real value
  const T *cs = conv_state + (b * Dn + c) * (k - 1);
  const float first = k > 1 ? to_f32(cs[0]) : to_f32(xp[b * xp_rs + c]);
  const float acc = w[c * k + (k - 1)] * first + bias[c];
  xc[b * Dn + c] = from_f32<T>(acc / (1.f + expf(-acc)));
  // new cache = the last k-1 tokens of the window (conv_state_out MAY be conv_state itself: a thread reads its channel's
  // window into registers before it writes - the captured decode graph updates its cache in place)
  T *co = conv_state_out + (b * Dn + c) * (k - 1);
  // (the host accepts k <= 16: k - 2 <= 14 shifted entries.  Round 4 kept three - right for the default k = 4, and for k >= 6
  //  left co[3 .. k-3] unwritten: stale in place, uninitialised out of place)
  constexpr int KEEP = 14;
  T keep[KEEP];
#pragma unroll
  for (int j = 0; j < KEEP; ++j) keep[j] = j + 1 < k - 1 ? cs[j + 1] : T(0);
  const T last = xp[b * xp_rs + c];
#pragma unroll
  for (int j = 0; j < KEEP; ++j)
    if (j + 1 < k - 1) co[j] = keep[j];
  if (k > 1) co[k - 2] = last;
}

template <typename T>
__global__ void __launch_bounds__(256)
decode_state_k(const float *__restrict__ dt_logits, const float *__restrict__ A_log, const T *__restrict__ Bt, int64_t bt_rs,
               const T *__restrict__ C, int64_t c_rs, const T *__restrict__ xc, const T *__restrict__ z, int64_t z_rs,
               const float *__restrict__ Dv, float *__restrict__ state, T *__restrict__ out, int64_t B, int64_t h, int64_t N,
               int softplus, const T *__restrict__ dt_in = nullptr, int64_t dt_rs = 0, const float *__restrict__ Wdt = nullptr,
               const float *__restrict__ bdt = nullptr, int R = 0) {
  const int64_t Dn = h * N;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= B * Dn) return;
  const int64_t b = i / Dn, c = i - b * Dn;
  float dlv;
  if (dt_in) {
    // dt_proj_head here (core.py:382; one launch less per layer of a token step): every channel of a head forms the head's
    // logit itself - apertis_tiny_linear_fwd's chain (bias first, then r = 0, 1, ...), the same bits
    const int64_t hd = c / N;
    const T *xr = dt_in + b * dt_rs;
    const float *w = Wdt + hd * R;
    dlv = bdt ? bdt[hd] : 0.f;
    for (int r = 0; r < R; ++r) dlv = fmaf(to_f32(xr[r]), w[r], dlv);
  } else {
    dlv = dt_logits[b * h + c / N];
  }
  if (softplus) dlv = softplus_f(dlv);
  const float av = __builtin_amdgcn_exp2f(dlv * (-expf(A_log[c]) * LOG2E_F));
  const float s = fmaf(av, state[b * Dn + c], to_f32(Bt[b * bt_rs + c]));
  state[b * Dn + c] = s;
  const float yv = to_f32(C[b * c_rs + c]) * s;
  const float dx = Dv[c] * to_f32(xc[b * Dn + c]);
  const float v = yv + dx;
  out[b * Dn + c] = from_f32<T>(v * silu_g(to_f32(z[b * z_rs + c])));
}


// ---------------------------------------------------------------------------------------------------------------
// LEAN forward (round 4; bf16, N = 16).  The kernels above stage tiles through LDS, walk LDS columns, hand y across a
// transpose and synchronise a 768-thread work-group five times per 64 tokens: ~700 instructions per wave and item, 58 % of a
// wave's life parked, two work-groups per CU (round 2-3 profiles) - their phases are a latency chain that nothing overlaps,
// and three restructurings of that design did not move it.  Here NOTHING is staged: a lane owns FOUR consecutive channels
// (8 bytes of a bf16 row; the Dn/4 lanes of a row cover it as one contiguous run), a wave owns one item of 64 tokens and
// walks it token by token with the loads of the next tokens in flight (two register batches of U tokens) - every operand
// (Bt, C, xc, z, out) is touched in its row-major place, there is no LDS, no barrier, no transpose, and a CU holds 11
// independent waves instead of two work-groups.  The carry between chunks comes from a state pass and a prefix kernel (three
// launches; Bt and delta are read twice - the second time from the Infinity Cache when they fit):
//   scan_lean_state_k   per item (P, S) = (prod a, state at the end from zero)       reads delta, Bt    writes agg
//   scan_lean_prefix_k  per (batch, four channels) the state entering every chunk     reads agg (, h0)   writes h_in
//   scan_lean_fwd_k     replay from h_in, skip + gate                                 reads delta, Bt, C, xc, z, h_in; writes out
// The composition order is fixed (token by token; the chunks by a fixed tree): run-to-run identical bits.
// Measured at B = 44, L = 4096, Dn = 176 (cold caches, rocprofv3): state 19.7 + prefix 6.7 + replay 59.4 us = 90 us end to end
// against 108-111 us for the single-pass kernel above on the same box (44.9 % against 36.6 % of the HBM peak); config 2 (Dn =
// 224, L = 2048, B = 32) 49 against 60 us.  What was tried on the way, each measured:
//   * the (item, four-channel group) pairs as one flat list, 64 consecutive entries per wave - every lane works (44 of 64 do
//     at Dn = 176), 31 % fewer wave-instructions, and SLOWER (replay 77 vs 72 us, state 24 vs 23): the kernels live on the
//     number of independent waves per CU (11 against 7.6), not on instruction issue;
//   * load batches of 2 / 4 / 8 tokens: replay 60.3 / 61.6 / 68.6 us (registers against waves per SIMD);
//   * the state pass's Bt loads non-temporal: the replay then re-reads Bt from HBM, 72 instead of 60 us;
//   * the prefix composed inside the replay (every item folds the records in front of it, 31 on average): no prefix launch,
//     but 125 MB more through L2 and up to eight dependent round trips before an item starts - replay 97 us, 117 end to end;
//   * a thread per channel walking its 64 records in the prefix kernel: 6.7 us, the same as the wave-parallel scan below - a
//     small kernel between two big ones costs its launch, one cold round trip and its drain whatever it computes.
// ---------------------------------------------------------------------------------------------------------------
#ifndef LEAN_U_
#define LEAN_U_ 4
#endif
#ifndef LEAN_US_
#define LEAN_US_ 4
#endif
#ifndef LEAN_NW_
#define LEAN_NW_ 4
#endif
constexpr int LEAN_U = LEAN_U_, LEAN_US = LEAN_US_;   // tokens per load batch (two batches in flight): replay, state pass
constexpr int LEAN_NW = LEAN_NW_;                     // waves per item in the state pass


// A wave takes floor(64 / g) whole items (g = Dn / 4 lanes each; ONE at Dn = 176, where 20 lanes idle - see above).
struct LeanItem { bool ok; int b, chunk, rows, c0, hh; uint32_t tok0; };
__device__ __forceinline__ LeanItem lean_item(const ScanDims &d, int g, int64_t items, bool reverse) {
  LeanItem it;
  const int ln = (int)threadIdx.x & 63, R = 64 / g, r = ln / g, q = ln - r * g;
  const int64_t item = (int64_t)blockIdx.x * R + r;
  it.ok = r < R && item < items;
  const int64_t bi = it.ok ? item / d.nchunks : 0;
  it.b = (int)bi;
  it.chunk = it.ok ? (int)(item - bi * d.nchunks) : 0;
  if (reverse) it.chunk = d.nchunks - 1 - it.chunk;
  const int64_t t0 = (int64_t)it.chunk * LTG;
  it.rows = it.ok ? (int)min((int64_t)LTG, d.L - t0) : 0;
  it.tok0 = (uint32_t)((int64_t)it.b * d.L + t0);
  it.c0 = 4 * q;
  it.hh = it.c0 >> d.log2N;
  return it;
}

// State pass.  NW waves share an item (LTG / NW consecutive tokens each); their aggregates meet in LDS and the last wave
// composes them in order.
// R8 > 0 (N4, round 4): dt_proj_head inside the pass - the delta logits are not read but formed from the dt columns of the
// projection output p (R <= 8 * R8 bf16 per token, the same padded row the pass streams Bt from: [Bt | 0 | C | 0 | dt | 0]) and
// the [h, R] weight + bias, and WRITTEN to dl for the replay and the backward: lane i of a head's quad takes token t0 + i as it
// does for the softplus, the weight rows sit in LDS (zero-padded to 8 * R8 columns, read four at a time, a quad shares its
// address) and the accumulation is apertis_tiny_linear_fwd's chain (bias first, then r = 0, 1, ... in order: the same bits).
template <int U, int NW, int R8 = 0>
__global__ void __launch_bounds__(64 * NW)
scan_lean_state_k(LeanT dl, const float *__restrict__ A_log, LeanT tb_, float2 *__restrict__ agg, ScanDims d, int g, int64_t items,
                  LeanT tdt = LeanT{nullptr, 0u, 0u}, const float *__restrict__ Wdt = nullptr, const float *__restrict__ bdt = nullptr,
                  int R = 0) {
  static_assert(U % 4 == 0 && (LTG / NW) % (2 * U) == 0, "delta comes in groups of four tokens; whole double batches per wave");
  constexpr int TW = LTG / NW;                                  // tokens per wave
  __shared__ float4 part[NW > 1 ? NW - 1 : 1][64][2];
  __shared__ __attribute__((aligned(16))) float sW[R8 > 0 ? 16 * (8 * R8 + 4) : 4];   // [head][8*R8 weights, bias, 3 pad]
  const int wv = (int)threadIdx.x >> 6;
  const LeanItem it = lean_item(d, g, items, false);
  float A2[4] = {0.f, 0.f, 0.f, 0.f}, S[4] = {0.f, 0.f, 0.f, 0.f}, sumdl = 0.f;
  if constexpr (R8 > 0) {
    constexpr int WP = 8 * R8 + 4;
    for (int i = (int)threadIdx.x; i < 16 * WP; i += 64 * NW) {
      const int j = i / WP, r = i - j * WP;
      sW[i] = j < d.h ? (r < R ? Wdt[j * R + r] : (r == 8 * R8 && bdt) ? bdt[j] : 0.f) : 0.f;
    }
    __syncthreads();
  }
  if (it.ok) {
#pragma unroll
    for (int k = 0; k < 4; ++k) A2[k] = -expf(A_log[it.c0 + k]) * LOG2E_F;
    const __amdgpu_buffer_rsrc_t rb = lean_rsrc(tb_.p, tb_.bytes), rd = lean_rsrc(dl.p, dl.bytes);
    const int qi = (int)(threadIdx.x & 3);                    // this lane's token inside a group of four
    const uint32_t tk0 = it.tok0 + (uint32_t)(wv * TW);
    const int rows = __builtin_amdgcn_readfirstlane(it.rows) - wv * TW;   // (<= 0: this wave's tokens are all past the end)
    const uint32_t ob = (tk0 * tb_.rs + (uint32_t)it.c0) * 2u, od = (tk0 * dl.rs + (uint32_t)it.hh) * 4u;
    uint2 vb[2][U];
    float vd[2][U / 4];
    typedef unsigned lean_u4 __attribute__((ext_vector_type(4)));
    [[maybe_unused]] lean_u4 vt[2][U / 4][R8 > 0 ? R8 : 1];
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rt = lean_rsrc(R8 > 0 ? tdt.p : dl.p, R8 > 0 ? tdt.bytes : 0u);
    [[maybe_unused]] const uint32_t ot = tk0 * tdt.rs * 2u;      // (the dt row of this wave's first token; the token rides in the lane offset)
    [[maybe_unused]] const float *swh = sW + it.hh * (8 * R8 + 4);
    const bool ragged = rows < TW;
    const int tlast = rows - 1;                                   // loads past the item's last row re-read that row (never used)
    auto ld = [&](int s, int tb) {
      if (rows <= 0) return;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        // (default cache policy: the replay reads these rows again, from the Infinity Cache where they fit)
        const lean_u2 t2 = __builtin_amdgcn_raw_buffer_load_b64(rb, (int)ob, (int)((uint32_t)min(tb + u, tlast) * tb_.rs * 2u), 0);
        vb[s][u] = make_uint2(t2[0], t2[1]);
      }
#pragma unroll
      for (int j = 0; j < U / 4; ++j) {  // (lane qi of a quad takes token tb + 4j + qi: the token rides in the lane offset here)
        if constexpr (R8 > 0) {
          const uint32_t o = ot + (uint32_t)min(tb + 4 * j + qi, tlast) * tdt.rs * 2u;
#pragma unroll
          for (int c = 0; c < R8; ++c) vt[s][j][c] = __builtin_amdgcn_raw_buffer_load_b128(rt, (int)(o + 16u * c), 0, 0);
        } else {
          vd[s][j] = lean_ld4(rd, od + (uint32_t)min(tb + 4 * j + qi, tlast) * dl.rs * 4u, 0u);
        }
      }
    };
    auto use = [&](int s, int tb) {
#pragma unroll
      for (int j = 0; j < U / 4; ++j) {
        if constexpr (R8 > 0) {
          float a = swh[8 * R8];
#pragma unroll
          for (int c = 0; c < R8; ++c) {
            const float4 w0 = *reinterpret_cast<const float4 *>(swh + 8 * c), w1 = *reinterpret_cast<const float4 *>(swh + 8 * c + 4);
            const lean_u4 x = vt[s][j][c];
            if (8 * c < R) {      // (whole chunks of four, like the stand-alone kernel: pad entries are zeros on both sides)
              a = fmaf(__uint_as_float(x[0] << 16), w0.x, a); a = fmaf(__uint_as_float(x[0] & 0xffff0000u), w0.y, a);
              a = fmaf(__uint_as_float(x[1] << 16), w0.z, a); a = fmaf(__uint_as_float(x[1] & 0xffff0000u), w0.w, a);
            }
            if (8 * c + 4 < R) {
              a = fmaf(__uint_as_float(x[2] << 16), w1.x, a); a = fmaf(__uint_as_float(x[2] & 0xffff0000u), w1.y, a);
              a = fmaf(__uint_as_float(x[3] << 16), w1.z, a); a = fmaf(__uint_as_float(x[3] & 0xffff0000u), w1.w, a);
            }
          }
          vd[s][j] = a;
          // the logits for the replay and the backward (tokens past the item's end: an offset the descriptor drops)
          const int t = tb + 4 * j + qi;
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a), rd, (int)(t < rows ? od + (uint32_t)t * dl.rs * 4u : 0xfffffff0u), 0, 0);
        }
        float sp = d.softplus ? softplus_fast(vd[s][j]) : vd[s][j];
        if (ragged) sp = tb + 4 * j + qi < rows ? sp : 0.f;       // a token past the item's end: delta 0 -> a = 1 ...
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int u = 4 * j + i;
          const float dlv = quad_bc(sp, i);
          float bv[4];
          unpack4(vb[s][u], bv);
          const bool v = !ragged || tb + u < rows;                  // ... and Bt 0: the state passes through
          sumdl += dlv;
#pragma unroll
          for (int k = 0; k < 4; ++k) S[k] = fmaf(__builtin_amdgcn_exp2f(dlv * A2[k]), S[k], v ? bv[k] : 0.f);
        }
      }
    };
    if (rows > 0) {
      ld(0, 0);
#pragma unroll 1
      for (int tb = 0; tb < TW; tb += 2 * U) {
        ld(1, tb + U);
        use(0, tb);
        ld(0, tb + 2 * U);
        use(1, tb + U);
      }
    }
  }
  // prod a_t = exp2(A2 * sum delta_t): one exp2 per channel instead of a multiply per token
  float P[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) P[k] = __builtin_amdgcn_exp2f(sumdl * A2[k]);
  if constexpr (NW > 1) {
    const int ln = (int)threadIdx.x & 63;
    if (wv < NW - 1) { part[wv][ln][0] = make_float4(P[0], S[0], P[1], S[1]); part[wv][ln][1] = make_float4(P[2], S[2], P[3], S[3]); }
    __syncthreads();
    if (wv != NW - 1) return;
    // (P, S) of the item = the waves' aggregates composed left to right; this (last) wave holds the rightmost
    float Pa[4] = {1.f, 1.f, 1.f, 1.f}, Sa[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w2 = 0; w2 < NW - 1; ++w2) {
      const float4 r0 = part[w2][ln][0], r1 = part[w2][ln][1];
      const float pw[4] = {r0.x, r0.z, r1.x, r1.z}, sw[4] = {r0.y, r0.w, r1.y, r1.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) { Sa[k] = fmaf(pw[k], Sa[k], sw[k]); Pa[k] *= pw[k]; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { S[k] = fmaf(P[k], Sa[k], S[k]); P[k] *= Pa[k]; }
  }
  if (!it.ok) return;
  float4 *o = reinterpret_cast<float4 *>(agg + ((int64_t)it.b * d.nchunks + it.chunk) * d.Dn + it.c0);
  o[0] = make_float4(P[0], S[0], P[1], S[1]);
  o[1] = make_float4(P[2], S[2], P[3], S[3]);
}

// State entering every chunk (saved for the backward as well): h_in[b][chunk][c].  One wave per (batch, four-channel group),
// lane = chunk: the records of (up to) 64 chunks are loaded in one round trip and composed by a log-step scan across the
// lanes; sequences of more than 64 chunks continue block by block with the carry of the previous block.
// REVERSE (backward): the chunks right to left - mu entering a chunk from the right, from the reverse aggregates.  (A template
// parameter so that the two uses have two kernel names in a trace.)
template <bool REVERSE>
__global__ void __launch_bounds__(64)
scan_lean_prefix_k(const float2 *__restrict__ agg, const float *__restrict__ h0, float *__restrict__ h_in, ScanDims d, int g) {
  constexpr bool reverse = REVERSE;
  const int64_t b = blockIdx.x / g;
  const int c0 = 4 * (int)(blockIdx.x - b * g), lane = (int)threadIdx.x;
  float hc[4] = {0.f, 0.f, 0.f, 0.f};
  if (h0) {
    const float4 t = *reinterpret_cast<const float4 *>(h0 + b * d.Dn + c0);
    hc[0] = t.x; hc[1] = t.y; hc[2] = t.z; hc[3] = t.w;
  }
  for (int j0 = 0; j0 < d.nchunks; j0 += 64) {
    const bool ok = j0 + lane < d.nchunks;
    const int j = reverse ? d.nchunks - 1 - (j0 + lane) : j0 + lane;
    float P[4] = {1.f, 1.f, 1.f, 1.f}, S[4] = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
      const float4 *r = reinterpret_cast<const float4 *>(agg + (b * d.nchunks + j) * d.Dn + c0);
      const float4 r0 = r[0], r1 = r[1];
      P[0] = r0.x; S[0] = r0.y; P[1] = r0.z; S[1] = r0.w; P[2] = r1.x; S[2] = r1.y; P[3] = r1.z; S[3] = r1.w;
    }
    // inclusive scan over the lanes: after the step of distance dd a lane holds the composite of the 2*dd records ending at it
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float pl = __shfl_up(P[k], dd), sl = __shfl_up(S[k], dd);
        if (lane >= dd) { S[k] = fmaf(P[k], sl, S[k]); P[k] *= pl; }
      }
    }
    // exclusive: the composite of the records before lane's, applied to the carry
    float hv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float pe = __shfl_up(P[k], 1), se = __shfl_up(S[k], 1);
      hv[k] = lane == 0 ? hc[k] : fmaf(pe, hc[k], se);
    }
    if (ok) *reinterpret_cast<float4 *>(h_in + (b * d.nchunks + j) * d.Dn + c0) = make_float4(hv[0], hv[1], hv[2], hv[3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) hc[k] = fmaf(__shfl(P[k], 63), hc[k], __shfl(S[k], 63));
  }
}

template <int U>
__global__ void __launch_bounds__(64)
scan_lean_fwd_k(LeanT dl, const float *__restrict__ A_log, LeanT tb_, LeanT tc, LeanT tx, LeanT tz, const float *__restrict__ Dv,
                const float *__restrict__ h_in, float *__restrict__ h_last, float *__restrict__ ckpt, LeanT to, ScanDims d, int g,
                int64_t items) {
  static_assert(U % 4 == 0, "delta comes in groups of four tokens");
  const LeanItem it = lean_item(d, g, items, false);
  if (!it.ok) return;
  float A2[4], Dk[4], hst[4];
  {
    const float4 hv = *reinterpret_cast<const float4 *>(h_in + ((int64_t)it.b * d.nchunks + it.chunk) * d.Dn + it.c0);
    hst[0] = hv.x; hst[1] = hv.y; hst[2] = hv.z; hst[3] = hv.w;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { A2[k] = -expf(A_log[it.c0 + k]) * LOG2E_F; Dk[k] = Dv[it.c0 + k]; }
  const __amdgpu_buffer_rsrc_t rb = lean_rsrc(tb_.p, tb_.bytes), rc = lean_rsrc(tc.p, tc.bytes), rx = lean_rsrc(tx.p, tx.bytes),
                               rz = lean_rsrc(tz.p, tz.bytes), rd = lean_rsrc(dl.p, dl.bytes);
  // (the output through a descriptor as well: a lane past its item's last row stores past the descriptor's end)
  const __amdgpu_buffer_rsrc_t ro = lean_rsrc(to.p, to.bytes);
  const int qi = (int)(threadIdx.x & 3);
  const uint32_t c2 = (uint32_t)it.c0 * 2u;
  const uint32_t ob = it.tok0 * tb_.rs * 2u + c2, oc = it.tok0 * tc.rs * 2u + c2, ox = it.tok0 * tx.rs * 2u + c2,
                 oz = it.tok0 * tz.rs * 2u + c2, oo = it.tok0 * to.rs * 2u + c2, od = (it.tok0 * dl.rs + (uint32_t)it.hh) * 4u;
  uint2 vb[2][U], vc[2][U], vx[2][U], vz[2][U];
  float vd[2][U / 4];
  const int rows_u = __builtin_amdgcn_readfirstlane(it.rows), tlast = rows_u - 1;
  const bool ragged = rows_u < LTG;
  auto ld = [&](int s, int tb) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t t = (uint32_t)min(tb + u, tlast);           // (past the item's last row: that row again, never used)
      const lean_u2 t2 = __builtin_amdgcn_raw_buffer_load_b64(rb, (int)ob, (int)(t * tb_.rs * 2u), 0);   // (the state pass left these in the caches)
      vb[s][u] = make_uint2(t2[0], t2[1]);
      vc[s][u] = lean_ld8(rc, oc, t * tc.rs * 2u);
      vx[s][u] = lean_ld8(rx, ox, t * tx.rs * 2u);
      vz[s][u] = lean_ld8(rz, oz, t * tz.rs * 2u);
    }
#pragma unroll
    for (int j = 0; j < U / 4; ++j) vd[s][j] = lean_ld4(rd, od + (uint32_t)min(tb + 4 * j + qi, tlast) * dl.rs * 4u, 0u);
  };
  auto use = [&](int s, int tb) {
#pragma unroll
    for (int j = 0; j < U / 4; ++j) {
      float sp = d.softplus ? softplus_fast(vd[s][j]) : vd[s][j];
      if (ragged) sp = tb + 4 * j + qi < it.rows ? sp : 0.f;
      // the state entering every fourth token, for the lean backward (which rebuilds four states at a time from it)
      if (ckpt && tb + 4 * j < it.rows)
        *reinterpret_cast<float4 *>(ckpt + ((int64_t)it.b * d.nck + (((int64_t)it.chunk * LTG + tb) >> 2) + j) * d.Dn + it.c0) =
            make_float4(hst[0], hst[1], hst[2], hst[3]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int u = 4 * j + i;
        const float dlv = quad_bc(sp, i);
        const bool v = !ragged || tb + u < it.rows;
        float bv[4], cv[4], xv[4], zv[4], o[4];
        unpack4(vb[s][u], bv); unpack4(vc[s][u], cv); unpack4(vx[s][u], xv); unpack4(vz[s][u], zv);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float a = __builtin_amdgcn_exp2f(dlv * A2[k]);
          hst[k] = fmaf(a, hst[k], v ? bv[k] : 0.f);
          const float y = cv[k] * hst[k];
          const float dx = Dk[k] * xv[k];
          const float val = y + dx;
          o[k] = val * silu_g(zv[k]);
        }
        const lean_u2 ov = {lean_pack2(o[0], o[1]), lean_pack2(o[2], o[3])};
        // rows past the item's end: an offset no descriptor covers (the store is dropped)
        __builtin_amdgcn_raw_buffer_store_b64(ov, ro, (int)(v ? oo : 0xfffffff0u), (int)((uint32_t)(tb + u) * to.rs * 2u), 2);
      }
    }
  };
  ld(0, 0);
#pragma unroll 1
  for (int tb = 0; tb < LTG; tb += 2 * U) {
    ld(1, tb + U);
    use(0, tb);
    ld(0, tb + 2 * U);
    use(1, tb + U);
  }
  if (h_last && it.chunk == d.nchunks - 1)
    *reinterpret_cast<float4 *>(h_last + (int64_t)it.b * d.Dn + it.c0) = make_float4(hst[0], hst[1], hst[2], hst[3]);
}

// ---------------------------------------------------------------------------------------------------------------
// LEAN backward: the same geometry (lane = four channels, wave = one 64-token item, no LDS staging), three launches:
//   scan_lean_bstate_k   per item the reverse aggregate (P, M): mu at the item's first token from zero at its end
//                        reads delta, C, dout, z                                        writes agg
//   scan_lean_prefix_k   (reverse) mu entering every chunk from the right               writes mu_in
//   scan_lean_bwd_k      per item, blocks of four tokens right to left: the four states of a block are rebuilt from the state
//                        entering it (saved by the lean forward: `ckpt`, every fourth token), then the adjoint
//                          lambda_t = dv_t C_t + mu_{t+1},  mu_t = a_t lambda_t,   dv = dout silu(z)
//                        emits dBt = lambda, dC = dv s, dxc = dv D, dz = dout silu'(z) (C s + D xc), d delta = sum over the head of
//                        lambda s_{t-1} a A, and the item's partial sums of dA_log and dD (folded by colsum_kernel as before)
//                        reads delta, Bt, C, xc, z, dout, ckpt, mu_in; writes dBt, dC, dxc, dz, d_delta, part
// ---------------------------------------------------------------------------------------------------------------

template <int NW>
__global__ void __launch_bounds__(64 * NW)
scan_lean_bstate_k(LeanT dl, const float *__restrict__ A_log, LeanT tc, LeanT tg, LeanT tz, float2 *__restrict__ agg, ScanDims d,
                   int g, int64_t items) {
  constexpr int TW = LTG / NW, U = 4;
  static_assert(TW % (2 * U) == 0, "whole double batches per wave");
  __shared__ float4 part[NW > 1 ? NW - 1 : 1][64][2];
  const int wv = (int)threadIdx.x >> 6;
  const LeanItem it = lean_item(d, g, items, false);
  float A2[4] = {0.f, 0.f, 0.f, 0.f}, M[4] = {0.f, 0.f, 0.f, 0.f}, sumdl = 0.f;
  if (it.ok) {
#pragma unroll
    for (int k = 0; k < 4; ++k) A2[k] = -expf(A_log[it.c0 + k]) * LOG2E_F;
    const __amdgpu_buffer_rsrc_t rc = lean_rsrc(tc.p, tc.bytes), rg = lean_rsrc(tg.p, tg.bytes), rz = lean_rsrc(tz.p, tz.bytes),
                                 rd = lean_rsrc(dl.p, dl.bytes);
    const int qi = (int)(threadIdx.x & 3);
    const uint32_t tk0 = it.tok0 + (uint32_t)(wv * TW);
    const int rows = __builtin_amdgcn_readfirstlane(it.rows) - wv * TW, tlast = rows - 1;
    const uint32_t c2 = (uint32_t)it.c0 * 2u;
    const uint32_t oc = tk0 * tc.rs * 2u + c2, og = tk0 * tg.rs * 2u + c2, oz = tk0 * tz.rs * 2u + c2,
                   od = (tk0 * dl.rs + (uint32_t)it.hh) * 4u;
    uint2 vc[2][U], vg[2][U], vz[2][U];
    float vd[2];
    const bool ragged = rows < TW;
    auto ld = [&](int s, int tb) {   // tokens tb .. tb+3 (default cache policy: the adjoint pass reads them again)
      if (tb < 0 || rows <= 0) return;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t t = (uint32_t)min(tb + u, tlast);
        const lean_u2 a2 = __builtin_amdgcn_raw_buffer_load_b64(rc, (int)oc, (int)(t * tc.rs * 2u), 0);
        const lean_u2 b2 = __builtin_amdgcn_raw_buffer_load_b64(rg, (int)og, (int)(t * tg.rs * 2u), 0);
        const lean_u2 c2_ = __builtin_amdgcn_raw_buffer_load_b64(rz, (int)oz, (int)(t * tz.rs * 2u), 0);
        vc[s][u] = make_uint2(a2[0], a2[1]); vg[s][u] = make_uint2(b2[0], b2[1]); vz[s][u] = make_uint2(c2_[0], c2_[1]);
      }
      vd[s] = lean_ld4(rd, od + (uint32_t)min(tb + qi, tlast) * dl.rs * 4u, 0u);
    };
    auto use = [&](int s, int tb) {
      if (tb < 0 || rows <= 0) return;
      float sp = d.softplus ? softplus_fast(vd[s]) : vd[s];
      if (ragged) sp = tb + qi < rows ? sp : 0.f;
#pragma unroll
      for (int i = 3; i >= 0; --i) {                                 // right to left
        const float dlv = quad_bc(sp, i);
        const bool v = !ragged || tb + i < rows;
        float cv[4], gv[4], zv[4];
        unpack4(vc[s][i], cv); unpack4(vg[s][i], gv); unpack4(vz[s][i], zv);
        sumdl += dlv;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float av = __builtin_amdgcn_exp2f(dlv * A2[k]);
          const float u = v ? (gv[k] * silu_g(zv[k])) * cv[k] : 0.f;
          M[k] = av * (u + M[k]);
        }
      }
    };
    ld(0, TW - U);
#pragma unroll 1
    for (int tb = TW - U; tb >= 0; tb -= 2 * U) {
      ld(1, tb - U);
      use(0, tb);
      ld(0, tb - 2 * U);
      use(1, tb - U);
    }
  }
  float P[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) P[k] = __builtin_amdgcn_exp2f(sumdl * A2[k]);
  if constexpr (NW > 1) {
    const int ln = (int)threadIdx.x & 63;
    if (wv > 0) { part[wv - 1][ln][0] = make_float4(P[0], M[0], P[1], M[1]); part[wv - 1][ln][1] = make_float4(P[2], M[2], P[3], M[3]); }
    __syncthreads();
    if (wv != 0) return;
    // the item's (P, M) = the waves' aggregates composed right to left; this (first) wave holds the leftmost
    float Pa[4] = {1.f, 1.f, 1.f, 1.f}, Ma[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w2 = NW - 1; w2 >= 1; --w2) {
      const float4 r0 = part[w2 - 1][ln][0], r1 = part[w2 - 1][ln][1];
      const float pw[4] = {r0.x, r0.z, r1.x, r1.z}, mw[4] = {r0.y, r0.w, r1.y, r1.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) { Ma[k] = fmaf(pw[k], Ma[k], mw[k]); Pa[k] *= pw[k]; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { M[k] = fmaf(P[k], Ma[k], M[k]); P[k] *= Pa[k]; }
  }
  if (!it.ok) return;
  float4 *o = reinterpret_cast<float4 *>(agg + ((int64_t)it.b * d.nchunks + it.chunk) * d.Dn + it.c0);
  o[0] = make_float4(P[0], M[0], P[1], M[1]);
  o[1] = make_float4(P[2], M[2], P[3], M[3]);
}

__global__ void __launch_bounds__(64)
scan_lean_bwd_k(LeanT dl, const float *__restrict__ A_log, LeanT tb_, LeanT tc, LeanT tx, LeanT tz, LeanT tg,
                const float *__restrict__ Dv, const float *__restrict__ ckpt, const float *__restrict__ mu_in, LeanT ob_, LeanT oc_,
                int store_w, LeanT ox_, LeanT oz_, float *__restrict__ d_dlt, float *__restrict__ part, ScanDims d, int g,
                int64_t items) {
  const LeanItem it = lean_item(d, g, items, false);
  if (!it.ok) return;
  float A2[4], Ac[4], Dk[4], mu[4], dA[4] = {0.f, 0.f, 0.f, 0.f}, dD[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) { Ac[k] = -expf(A_log[it.c0 + k]); A2[k] = Ac[k] * LOG2E_F; Dk[k] = Dv[it.c0 + k]; }
  {
    const float4 m = *reinterpret_cast<const float4 *>(mu_in + ((int64_t)it.b * d.nchunks + it.chunk) * d.Dn + it.c0);
    mu[0] = m.x; mu[1] = m.y; mu[2] = m.z; mu[3] = m.w;
  }
  const __amdgpu_buffer_rsrc_t rb = lean_rsrc(tb_.p, tb_.bytes), rc = lean_rsrc(tc.p, tc.bytes), rx = lean_rsrc(tx.p, tx.bytes),
                               rz = lean_rsrc(tz.p, tz.bytes), rg = lean_rsrc(tg.p, tg.bytes), rd = lean_rsrc(dl.p, dl.bytes);
  const __amdgpu_buffer_rsrc_t wb = lean_rsrc(ob_.p, ob_.bytes), wc = lean_rsrc(oc_.p, oc_.bytes), wx = lean_rsrc(ox_.p, ox_.bytes),
                               wz = lean_rsrc(oz_.p, oz_.bytes);
  const int qi = (int)(threadIdx.x & 3);
  const uint32_t c2 = (uint32_t)it.c0 * 2u;
  const uint32_t fb = it.tok0 * tb_.rs * 2u + c2, fc = it.tok0 * tc.rs * 2u + c2, fx = it.tok0 * tx.rs * 2u + c2,
                 fz = it.tok0 * tz.rs * 2u + c2, fg = it.tok0 * tg.rs * 2u + c2, fd = (it.tok0 * dl.rs + (uint32_t)it.hh) * 4u;
  const uint32_t sb = it.tok0 * ob_.rs * 2u + c2, sc = it.tok0 * oc_.rs * 2u + c2, sx = it.tok0 * ox_.rs * 2u + c2,
                 sz = it.tok0 * oz_.rs * 2u + c2;
  // the pad columns [Dn, store_w) of dBt / dC receive zeros (the padded slices of the projection output's gradient): the
  // first (store_w - Dn) / 4 lanes of the item write them
  const int npad = (store_w - (int)d.Dn) >> 2, q = it.c0 >> 2;
  const uint32_t pz = (uint32_t)(d.Dn + 4 * q) * 2u;
  const float *ck = ckpt + ((int64_t)it.b * d.nck + (((int64_t)it.chunk * LTG) >> 2)) * d.Dn + it.c0;
  float *ddp = d_dlt + ((int64_t)it.tok0 + qi) * d.h + it.hh;   // (+ tb * h per block)
  const int rows_u = __builtin_amdgcn_readfirstlane(it.rows), tlast = rows_u - 1;
  const bool ragged = rows_u < LTG;
  constexpr int U = 4;
  uint2 vb[2][U], vc[2][U], vx[2][U], vz[2][U], vg[2][U];
  float vd[2];
  float4 vk[2];
  auto ld = [&](int s, int tb) {
    if (tb < 0 || tb >= rows_u) return;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t t = (uint32_t)min(tb + u, tlast);
      vb[s][u] = lean_ld8(rb, fb, t * tb_.rs * 2u);
      vc[s][u] = lean_ld8(rc, fc, t * tc.rs * 2u);
      vx[s][u] = lean_ld8(rx, fx, t * tx.rs * 2u);
      vz[s][u] = lean_ld8(rz, fz, t * tz.rs * 2u);
      vg[s][u] = lean_ld8(rg, fg, t * tg.rs * 2u);
    }
    vd[s] = lean_ld4(rd, fd + (uint32_t)min(tb + qi, tlast) * dl.rs * 4u, 0u);
    vk[s] = *reinterpret_cast<const float4 *>(ck + (int64_t)(tb >> 2) * d.Dn);
  };
  auto use = [&](int s, int tb) {
    if (tb < 0 || tb >= rows_u) return;
    const float spx = d.softplus ? softplus_fast(vd[s]) : vd[s];
    const float sp = (ragged && tb + qi >= it.rows) ? 0.f : spx;
    // the four states of the block from the state entering it
    float hs[4][4], hin[4] = {vk[s].x, vk[s].y, vk[s].z, vk[s].w}, dlv[4];
    {
      float hc[4] = {hin[0], hin[1], hin[2], hin[3]};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        dlv[i] = quad_bc(sp, i);
        float bv[4];
        unpack4(vb[s][i], bv);
#pragma unroll
        for (int k = 0; k < 4; ++k) { hc[k] = fmaf(__builtin_amdgcn_exp2f(dlv[i] * A2[k]), hc[k], bv[k]); hs[i][k] = hc[k]; }
      }
    }
    float ddl_keep = 0.f;
#pragma unroll
    for (int i = 3; i >= 0; --i) {
      const bool v = !ragged || tb + i < it.rows;
      float cv[4], xv[4], zv[4], gv[4], oB[4], oC[4], oX[4], oZ[4];
      unpack4(vc[s][i], cv); unpack4(vx[s][i], xv); unpack4(vz[s][i], zv); unpack4(vg[s][i], gv);
      float qs = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float av = __builtin_amdgcn_exp2f(dlv[i] * A2[k]);
        float f, df;
        silu_both(zv[k], f, df);
        const float dv = gv[k] * f;
        const float lam = fmaf(dv, cv[k], mu[k]);
        const float hprev = i > 0 ? hs[i - 1][k] : hin[k];
        const float qq = lam * hprev * av * Ac[k];                 // da_t * a_t * A
        const float y = cv[k] * hs[i][k];
        const float dx = Dk[k] * xv[k];
        const float val = y + dx;
        oB[k] = lam;
        oC[k] = dv * hs[i][k];
        oX[k] = dv * Dk[k];
        oZ[k] = gv[k] * df * val;
        if (v) {
          dA[k] = fmaf(qq, dlv[i], dA[k]);
          dD[k] = fmaf(dv, xv[k], dD[k]);
          qs += qq;
          mu[k] = av * lam;
        }
      }
      const uint32_t t = (uint32_t)(tb + i);
      const uint32_t bad = 0xfffffff0u;
      const lean_u2 wB = {lean_pack2(oB[0], oB[1]), lean_pack2(oB[2], oB[3])}, wC = {lean_pack2(oC[0], oC[1]), lean_pack2(oC[2], oC[3])},
                    wX = {lean_pack2(oX[0], oX[1]), lean_pack2(oX[2], oX[3])}, wZ = {lean_pack2(oZ[0], oZ[1]), lean_pack2(oZ[2], oZ[3])};
      __builtin_amdgcn_raw_buffer_store_b64(wB, wb, (int)(v ? sb : bad), (int)(t * ob_.rs * 2u), 2);
      __builtin_amdgcn_raw_buffer_store_b64(wC, wc, (int)(v ? sc : bad), (int)(t * oc_.rs * 2u), 2);
      __builtin_amdgcn_raw_buffer_store_b64(wX, wx, (int)(v ? sx : bad), (int)(t * ox_.rs * 2u), 2);
      __builtin_amdgcn_raw_buffer_store_b64(wZ, wz, (int)(v ? sz : bad), (int)(t * oz_.rs * 2u), 2);
      if (q < npad) {
        const lean_u2 zz = {0u, 0u};
        __builtin_amdgcn_raw_buffer_store_b64(zz, wb, (int)(v ? it.tok0 * ob_.rs * 2u + pz : bad), (int)(t * ob_.rs * 2u), 2);
        __builtin_amdgcn_raw_buffer_store_b64(zz, wc, (int)(v ? it.tok0 * oc_.rs * 2u + pz : bad), (int)(t * oc_.rs * 2u), 2);
      }
      // d delta of the head: the quad's 16 channels; token tb+i's value parks in lane i of the quad, one store per block
      float dq = quad_sum(qs);
      if (d.softplus) dq *= 1.f - __builtin_amdgcn_exp2f(-dlv[i] * LOG2E_F);      // sigmoid(x) = 1 - exp(-softplus(x))
      if (qi == i) ddl_keep = dq;
    }
    if (tb + qi < it.rows) ddp[(int64_t)tb * d.h] = ddl_keep;
  };
#ifndef LEAN_BWD_DB
#define LEAN_BWD_DB 0
#endif
#if LEAN_BWD_DB   // two blocks' loads in flight: 209 VGPRs = two waves per SIMD = 8 of a CU's 11 items resident (measured: 133 us)
  ld(0, LTG - U);
#pragma unroll 1
  for (int tb = LTG - U; tb >= 0; tb -= 2 * U) {
    ld(1, tb - U);
    use(0, tb);
    ld(0, tb - 2 * U);
    use(1, tb - U);
  }
#else             // one block at a time, three waves per SIMD: every item of a CU resident at once, the waves cover each other
#pragma unroll 1
  for (int tb = LTG - U; tb >= 0; tb -= U) {
    ld(0, tb);
    use(0, tb);
  }
#endif
  float4 *po = reinterpret_cast<float4 *>(part + (((int64_t)it.b * d.nchunks + it.chunk) * 2) * d.Dn + it.c0);
  po[0] = make_float4(dA[0], dA[1], dA[2], dA[3]);
  *reinterpret_cast<float4 *>(part + (((int64_t)it.b * d.nchunks + it.chunk) * 2 + 1) * d.Dn + it.c0) = make_float4(dD[0], dD[1], dD[2], dD[3]);
}

// lanes per item (g); false when the lean kernels do not take the shape
static inline bool lean_geometry(const ScanDims &d, int &g) {
  // N = 16: the four lanes of a head are a DPP quad.  128 < Dn <= 256: ONE item per wave, so the item's row count is
  // wave-uniform and the token index of a load can be clamped to the item's last row in a scalar register (the hardware range
  // check covers a lane's offset, not the scalar row term: an unclamped prefetch past the tensor's last row would read
  // unowned memory).  Narrower models (Dn = 64) stay on the staged kernels, which are faster there anyway (32 vs 35 us).
  if (d.N != 16 || d.Dn > 256 || d.Dn <= 128) return false;
  g = (int)(d.Dn / 4);
  return true;
}

// ---------------------------------------------------------------------------------------------------------------
// channel groups per work-group: as many as cover Dn (whole rows), bounded by what the LDS tiles allow
template <typename T> constexpr int cw_max(bool bwd) { return sizeof(T) == 2 ? (bwd ? 3 : 4) : 2; }
template <typename T> void pick_cw(int64_t Dn, bool bwd, int &cw, int &ncs) {
  const int g = (int)ceil_div64(Dn, TC);
  ncs = (g + cw_max<T>(bwd) - 1) / cw_max<T>(bwd);
  cw = (g + ncs - 1) / ncs;
}

template <typename T, int CW> size_t fwd_lds(const ScanDims &d) {
  typedef Geo<CW> G;
  return 2 * (size_t)LTG * G::CWC * sizeof(T) + (size_t)LTG * CW * d.HT * 4 + (G::NS + 2) * G::CWC * sizeof(float2) + G::CWC * 4 + 16;
}
template <typename T, int CW> size_t bwd_lds(const ScanDims &d) {
  typedef Geo<CW> G;
  return 2 * (size_t)LTG * G::CWC * sizeof(T) + (size_t)LTG * G::CWC * 4 + 2 * (size_t)LTG * CW * d.HT * 4 + G::NS * G::CWC * 12 +
         (G::NS + 2) * G::CWC * sizeof(float2) + G::CWC * 4 + 16;
}

template <typename F> void allow_lds(F *fn, size_t bytes) {
  if (bytes > 64 * 1024) (void)hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

struct FwdArgs {
  const float *dlt, *A_log; const void *Bt; int64_t bt_rs; const void *C; int64_t c_rs; const void *xc; int64_t xc_rs;
  const void *z; int64_t z_rs; const float *Dv, *h0; void *out; int64_t out_rs; float *h_last, *agg, *h_in; void *ws;
  uint32_t epoch; ScanDims d; int single_pass; hipStream_t st;
};

// minimum waves per SIMD the forward is compiled for: bf16 two work-groups per CU (four of the 512-thread ones)
template <typename T, int CW> constexpr int fwd_minw() {
  return sizeof(T) == 2 ? (CW == 1 ? 8 : CW == 2 ? 8 : CW == 3 ? 6 : 4) : 4;
}

template <typename T, int VB, int CW>
int launch_gate_fwd_cw(const FwdArgs &a, int ncs) {
  typedef Geo<CW> G;
  constexpr int MW = fwd_minw<T, CW>();
  const ScanDims &d = a.d;
  const size_t lds = fwd_lds<T, CW>(d);
  GateWsHead *head = reinterpret_cast<GateWsHead *>(a.ws);
  gran_t *gran = reinterpret_cast<gran_t *>(reinterpret_cast<char *>(a.ws) + sizeof(GateWsHead));
#define FWD_ARGS(aggp) a.dlt, a.A_log, (const T *)a.Bt, a.bt_rs, (const T *)a.C, a.c_rs, (const T *)a.xc, a.xc_rs, (const T *)a.z, \
    a.z_rs, a.Dv, a.h0, (float2 *)(aggp), head, gran, a.epoch, a.h_in, a.h_last, (T *)a.out, a.out_rs, d, ncs, (int)items
  const int64_t items = (int64_t)d.nchunks * d.B * ncs;
  if (items > 0x3fffffffLL) return APERTIS_ERR_UNSUPPORTED;
  dim3 grid((unsigned)items);                                       // one item per work-group, chunk-major
  if (a.single_pass) {
    allow_lds(scan_gate_fwd_k<T, VB, CW, 1, MW>, lds);
    hipLaunchKernelGGL((scan_gate_fwd_k<T, VB, CW, 1, MW>), grid, dim3(G::NTH), lds, a.st, FWD_ARGS(nullptr));
  } else {
    allow_lds(scan_gate_fwd_k<T, VB, CW, 2, MW>, lds);
    hipLaunchKernelGGL((scan_gate_fwd_k<T, VB, CW, 2, MW>), grid, dim3(G::NTH), lds, a.st, FWD_ARGS(a.agg));
    allow_lds(scan_gate_fwd_k<T, VB, CW, 0, MW>, lds);
    hipLaunchKernelGGL((scan_gate_fwd_k<T, VB, CW, 0, MW>), grid, dim3(G::NTH), lds, a.st, FWD_ARGS(a.agg));
  }
#undef FWD_ARGS
  return apertis_check_launch();
}

template <typename T, int VB> int launch_gate_fwd(const FwdArgs &a) {
  int cw, ncs;
  pick_cw<T>(a.d.Dn, false, cw, ncs);
  if (ncs > 65535) return APERTIS_ERR_UNSUPPORTED;
  if (cw == 1) return launch_gate_fwd_cw<T, VB, 1>(a, ncs);
  if (cw == 2) return launch_gate_fwd_cw<T, VB, 2>(a, ncs);
  if constexpr (sizeof(T) == 2) {
    if (cw == 3) return launch_gate_fwd_cw<T, VB, 3>(a, ncs);
    return launch_gate_fwd_cw<T, VB, 4>(a, ncs);
  }
  return APERTIS_ERR_UNSUPPORTED;
}

struct BwdArgs {
  const float *dlt, *A_log; const void *Bt; int64_t bt_rs; const void *C; int64_t c_rs; const void *xc; int64_t xc_rs;
  const void *z; int64_t z_rs; const float *Dv; const void *dout; int64_t do_rs; const float *h_in; void *dBt; int64_t dbt_rs;
  void *dC; int64_t dc_rs, store_w; void *dxc; int64_t dxc_rs; void *dz; int64_t dz_rs; float *d_dlt, *dA_dD, *agg, *fold, *part;
  void *ws; uint32_t epoch; ScanDims d; int single_pass; hipStream_t st;
};

template <typename T, int VB, int CW>
int launch_gate_bwd_cw(const BwdArgs &a, int ncs) {
  typedef Geo<CW> G;
  const ScanDims &d = a.d;
  const size_t lds = bwd_lds<T, CW>(d);
  GateWsHead *head = reinterpret_cast<GateWsHead *>(a.ws);
  gran_t *gran = reinterpret_cast<gran_t *>(reinterpret_cast<char *>(a.ws) + sizeof(GateWsHead));
#define BWD_ARGS(aggp) a.dlt, a.A_log, (const T *)a.Bt, a.bt_rs, (const T *)a.C, a.c_rs, (const T *)a.xc, a.xc_rs, (const T *)a.z, \
    a.z_rs, a.Dv, (const T *)a.dout, a.do_rs, a.h_in, (float2 *)(aggp), head, gran, a.epoch, (T *)a.dBt, a.dbt_rs, (T *)a.dC, a.dc_rs, \
    a.store_w, (T *)a.dxc, a.dxc_rs, (T *)a.dz, a.dz_rs, a.d_dlt, a.part, d, ncs
  if (a.single_pass) {
    const int64_t items = (int64_t)d.nchunks * d.B * ncs;
    if (items > 0x7fffffffLL) return APERTIS_ERR_UNSUPPORTED;
    allow_lds(scan_gate_bwd_k<T, VB, CW, 1>, lds);
    hipLaunchKernelGGL((scan_gate_bwd_k<T, VB, CW, 1>), dim3((unsigned)items), dim3(G::NTH), lds, a.st, BWD_ARGS(nullptr));
  } else {
    dim3 grid(d.nchunks, (unsigned)ncs, (unsigned)d.B);
    allow_lds(scan_gate_bwd_k<T, VB, CW, 2>, lds);
    hipLaunchKernelGGL((scan_gate_bwd_k<T, VB, CW, 2>), grid, dim3(G::NTH), lds, a.st, BWD_ARGS(a.agg));
    allow_lds(scan_gate_bwd_k<T, VB, CW, 0>, lds);
    hipLaunchKernelGGL((scan_gate_bwd_k<T, VB, CW, 0>), grid, dim3(G::NTH), lds, a.st, BWD_ARGS(a.agg));
  }
#undef BWD_ARGS
  // fold the per-chunk partials [rows][2*Dn] (dA_log | dD) in a fixed order, two levels
  const int64_t rows = d.B * d.nchunks, cols = 2 * d.Dn;
  const unsigned ctl = (unsigned)ceil_div64(cols, TC);
  if (rows <= 128) {
    hipLaunchKernelGGL(colsum_kernel, dim3(ctl), dim3(1024), 0, a.st, a.part, a.dA_dD, rows, cols, rows);
  } else {
    const int64_t groups = std::min<int64_t>(64, ceil_div64(rows, 64)), rpg = ceil_div64(rows, groups);
    const int64_t ng = ceil_div64(rows, rpg);
    hipLaunchKernelGGL(colsum_kernel, dim3(ctl, (unsigned)ng), dim3(1024), 0, a.st, a.part, a.fold, rows, cols, rpg);
    hipLaunchKernelGGL(colsum_kernel, dim3(ctl), dim3(1024), 0, a.st, a.fold, a.dA_dD, ng, cols, ng);
  }
  return apertis_check_launch();
}

template <typename T, int VB> int launch_gate_bwd(const BwdArgs &a) {
  int cw, ncs;
  pick_cw<T>(a.d.Dn, true, cw, ncs);
  if (ncs > 65535) return APERTIS_ERR_UNSUPPORTED;
  if (cw == 1) return launch_gate_bwd_cw<T, VB, 1>(a, ncs);
  if (cw == 2) return launch_gate_bwd_cw<T, VB, 2>(a, ncs);
  if constexpr (sizeof(T) == 2) return launch_gate_bwd_cw<T, VB, 3>(a, ncs);
  return APERTIS_ERR_UNSUPPORTED;
}

// widest access every slice allows: pointer, row stride and row length multiples of it (tile steps are multiples of 128 B)
template <typename T> int gate_align(std::initializer_list<std::pair<const void *, int64_t>> slices, int64_t width) {
  int al = 16;
  for (auto &s : slices)
    al = std::min(al, common_align({(uint64_t)(uintptr_t)s.first, (uint64_t)s.second * sizeof(T), (uint64_t)width * sizeof(T)}));
  return al;
}

}  // namespace

extern "C" int64_t apertis_scan_gate_chunk_len(void) { return LTG; }

extern "C" int64_t apertis_scan_gate_workspace_bytes(int64_t B, int64_t L, int64_t Dn) {
  // 64-byte head (two ticket counters, error word) + per (batch, channel super-tile): one record per chunk of 64 tokens and
  // one per super-chunk of 8 chunks, each 64*CW lanes x 2 granules x 8 bytes; sized for the padded channel count
  const int64_t nch = ceil_div64(L, LTG), nsup = ceil_div64(nch, SUP), g = ceil_div64(Dn, TC);
  return (int64_t)sizeof(GateWsHead) + B * (nch + nsup) * (g + 3) * TC * 2 * 8;
}

extern "C" int apertis_scan_gate_fwd(const float *dlt, const float *A_log, const void *Bt, int64_t bt_rs, const void *C,
                                     int64_t c_rs, const void *xc, int64_t xc_rs, const void *z, int64_t z_rs,
                                     const float *D, const float *h0, void *out, int64_t out_rs, float *h_last, float *agg,
                                     float *h_in, void *ws, uint32_t epoch, int64_t B, int64_t L, int64_t h, int64_t N,
                                     int dtype, int delta_softplus, int single_pass, void *stream) {
  if (!dlt || !A_log || !Bt || !C || !xc || !z || !D || !out || !h_in) return APERTIS_ERR_ARG;
  if (single_pass ? (!ws || epoch == 0) : !agg) return APERTIS_ERR_ARG;
  FwdArgs a{dlt, A_log, Bt, bt_rs, C, c_rs, xc, xc_rs, z, z_rs, D, h0, out, out_rs, h_last, agg, h_in, ws, epoch, {}, single_pass,
            (hipStream_t)stream};
  int rc = make_dims(a.d, B, L, h, N, delta_softplus);
  if (rc) return rc;
  a.d.nchunks = (int)ceil_div64(L, LTG);
  const int64_t Dn = a.d.Dn;
  if (bt_rs < Dn || c_rs < Dn || xc_rs < Dn || z_rs < Dn || out_rs < Dn) return APERTIS_ERR_ARG;
  if (dtype == APERTIS_F32) {
    const int al = gate_align<float>({{Bt, bt_rs}, {C, c_rs}, {xc, xc_rs}, {z, z_rs}, {out, out_rs}}, Dn);
    if (al >= 16) return launch_gate_fwd<float, 16>(a);
    if (al >= 8) return launch_gate_fwd<float, 8>(a);
    return launch_gate_fwd<float, 4>(a);
  } else if (dtype == APERTIS_BF16) {
    const int al = gate_align<bf16_t>({{Bt, bt_rs}, {C, c_rs}, {xc, xc_rs}, {z, z_rs}, {out, out_rs}}, Dn);
    if (al >= 16) return launch_gate_fwd<bf16_t, 16>(a);
    if (al >= 8) return launch_gate_fwd<bf16_t, 8>(a);
    if (al >= 4) return launch_gate_fwd<bf16_t, 4>(a);
    return launch_gate_fwd<bf16_t, 2>(a);
  }
  return APERTIS_ERR_UNSUPPORTED;
}

extern "C" int apertis_scan_gate_bwd(const float *dlt, const float *A_log, const void *Bt, int64_t bt_rs, const void *C,
                                     int64_t c_rs, const void *xc, int64_t xc_rs, const void *z, int64_t z_rs,
                                     const float *D, const void *dout, int64_t dout_rs, const float *h_in, void *dBt,
                                     int64_t dbt_rs, void *dC, int64_t dc_rs, int64_t store_w, void *dxc, int64_t dxc_rs,
                                     void *dz, int64_t dz_rs, float *d_dlt, float *dA_dD, float *agg, float *fold,
                                     float *part, void *ws, uint32_t epoch, int64_t B, int64_t L, int64_t h, int64_t N,
                                     int dtype, int delta_softplus, int single_pass, void *stream) {
  if (!dlt || !A_log || !Bt || !C || !xc || !z || !D || !dout || !h_in || !dBt || !dC || !dxc || !dz || !d_dlt || !dA_dD ||
      !fold || !part)
    return APERTIS_ERR_ARG;
  if (single_pass ? (!ws || epoch == 0) : !agg) return APERTIS_ERR_ARG;
  BwdArgs a{dlt, A_log, Bt, bt_rs, C, c_rs, xc, xc_rs, z, z_rs, D, dout, dout_rs, h_in, dBt, dbt_rs, dC, dc_rs, store_w, dxc,
            dxc_rs, dz, dz_rs, d_dlt, dA_dD, agg, fold, part, ws, epoch, {}, single_pass, (hipStream_t)stream};
  int rc = make_dims(a.d, B, L, h, N, delta_softplus);
  if (rc) return rc;
  a.d.nchunks = (int)ceil_div64(L, LTG);
  const int64_t Dn = a.d.Dn;
  if (bt_rs < Dn || c_rs < Dn || xc_rs < Dn || z_rs < Dn || dout_rs < Dn || dxc_rs < Dn || dz_rs < Dn) return APERTIS_ERR_ARG;
  // dBt / dC may be zero-extended up to the next multiple of 64 channels (the padded slices of the projection output)
  if (store_w < Dn || store_w > ceil_div64(Dn, TC) * TC || dbt_rs < store_w || dc_rs < store_w) return APERTIS_ERR_ARG;
  if (dtype == APERTIS_F32) {
    int al = gate_align<float>({{Bt, bt_rs}, {C, c_rs}, {xc, xc_rs}, {z, z_rs}, {dout, dout_rs}, {dxc, dxc_rs}, {dz, dz_rs}}, Dn);
    al = std::min(al, gate_align<float>({{dBt, dbt_rs}, {dC, dc_rs}}, store_w));
    if (al >= 16) return launch_gate_bwd<float, 16>(a);
    if (al >= 8) return launch_gate_bwd<float, 8>(a);
    return launch_gate_bwd<float, 4>(a);
  } else if (dtype == APERTIS_BF16) {
    int al = gate_align<bf16_t>({{Bt, bt_rs}, {C, c_rs}, {xc, xc_rs}, {z, z_rs}, {dout, dout_rs}, {dxc, dxc_rs}, {dz, dz_rs}}, Dn);
    al = std::min(al, gate_align<bf16_t>({{dBt, dbt_rs}, {dC, dc_rs}}, store_w));
    if (al >= 16) return launch_gate_bwd<bf16_t, 16>(a);
    if (al >= 8) return launch_gate_bwd<bf16_t, 8>(a);
    if (al >= 4) return launch_gate_bwd<bf16_t, 4>(a);
    return launch_gate_bwd<bf16_t, 2>(a);
  }
  return APERTIS_ERR_UNSUPPORTED;
}

// ---- lean forms (bf16, N = 16, 128 < Dn <= 256, 8-byte aligned slices, tensors below 4 GiB): APERTIS_ERR_UNSUPPORTED otherwise,
// the caller then takes apertis_scan_gate_fwd / _bwd ----
namespace {
struct LeanShape { ScanDims d; int g; int64_t T, items; };
int lean_shape(LeanShape &s, int64_t B, int64_t L, int64_t h, int64_t N, int softplus,
               std::initializer_list<std::pair<const void *, int64_t>> slices, int64_t width_max) {
  int rc = make_dims(s.d, B, L, h, N, softplus);
  if (rc) return rc;
  s.d.nchunks = (int)ceil_div64(L, LTG);
  if (!lean_geometry(s.d, s.g)) return APERTIS_ERR_UNSUPPORTED;
  s.T = B * L;
  s.items = B * s.d.nchunks;
  int64_t rs_max = 0;
  for (auto &sl : slices) {
    if (sl.second < s.d.Dn) return APERTIS_ERR_ARG;
    if ((((uintptr_t)sl.first) | (uint64_t)(sl.second * 2)) & 7) return APERTIS_ERR_UNSUPPORTED;
    rs_max = std::max(rs_max, sl.second);
  }
  if ((s.T * rs_max + width_max) * 2 >= 0xfff00000LL || s.T * h * 4 >= 0xfff00000LL) return APERTIS_ERR_UNSUPPORTED;
  return APERTIS_OK;
}
}  // namespace

static int lean_fwd_impl(const float *dlt, const void *dt_in, int64_t dt_rs, const float *W_dt, const float *b_dt, int64_t R,
                         const float *A_log, const void *Bt, int64_t bt_rs, const void *C, int64_t c_rs,
                         const void *xc, int64_t xc_rs, const void *z, int64_t z_rs, const float *D, const float *h0,
                         void *out, int64_t out_rs, float *h_last, float *agg, float *h_in, float *ckpt, int64_t B,
                         int64_t L, int64_t h, int64_t N, int delta_softplus, void *stream) {
  if (!dlt || !A_log || !Bt || !C || !xc || !z || !D || !out || !h_in || !agg) return APERTIS_ERR_ARG;
  LeanShape s;
  int rc = lean_shape(s, B, L, h, N, delta_softplus, {{Bt, bt_rs}, {C, c_rs}, {xc, xc_rs}, {z, z_rs}, {out, out_rs}}, h * N);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int64_t Dn = s.d.Dn, T = s.T;
  auto lt = [&](const void *p, int64_t rs) { return LeanT{p, (uint32_t)rs, (uint32_t)(((T - 1) * rs + Dn) * 2)}; };
  const LeanT tdl{dlt, (uint32_t)h, (uint32_t)(T * h * 4)};
  const unsigned grid = (unsigned)s.items;
  if (dt_in) {
    // dt_proj_head inside the state pass: whole 16-byte chunks of the dt columns (the caller's pad columns behind R are zeros
    // or at least finite: their weights are zeros here), at most 16 heads, the row reachable through a 32-bit descriptor
    const int64_t r8 = (R + 7) / 8;
    if (!W_dt || R < 1 || R > 64 || h > 16) return APERTIS_ERR_ARG;
    if (((uintptr_t)dt_in & 15) || dt_rs % 8 || dt_rs < 8 * r8 || ((T - 1) * dt_rs + 8 * r8) * 2 >= 0xFFFFFFF0LL) return APERTIS_ERR_UNSUPPORTED;
    const LeanT tdt{dt_in, (uint32_t)dt_rs, (uint32_t)(((T - 1) * dt_rs + 8 * r8) * 2)};
#define GO_DT(R8) hipLaunchKernelGGL((scan_lean_state_k<LEAN_US, LEAN_NW, R8>), dim3(grid), dim3(64 * LEAN_NW), 0, st, tdl, A_log, \
                                     lt(Bt, bt_rs), (float2 *)agg, s.d, s.g, s.items, tdt, W_dt, b_dt, (int)R)
    switch (r8) {
      case 1: GO_DT(1); break; case 2: GO_DT(2); break; case 3: GO_DT(3); break; case 4: GO_DT(4); break;
      case 5: GO_DT(5); break; case 6: GO_DT(6); break; case 7: GO_DT(7); break; default: GO_DT(8); break;
    }
#undef GO_DT
  } else {
    hipLaunchKernelGGL((scan_lean_state_k<LEAN_US, LEAN_NW>), dim3(grid), dim3(64 * LEAN_NW), 0, st, tdl, A_log, lt(Bt, bt_rs),
                       (float2 *)agg, s.d, s.g, s.items, LeanT{nullptr, 0u, 0u}, (const float *)nullptr, (const float *)nullptr, 0);
  }
  hipLaunchKernelGGL(scan_lean_prefix_k<false>, dim3((unsigned)(B * s.g)), dim3(64), 0, st, (const float2 *)agg, h0, h_in, s.d, s.g);
  hipLaunchKernelGGL(scan_lean_fwd_k<LEAN_U>, dim3(grid), dim3(64), 0, st, tdl, A_log, lt(Bt, bt_rs), lt(C, c_rs), lt(xc, xc_rs),
                     lt(z, z_rs), D, h_in, h_last, ckpt, lt(out, out_rs), s.d, s.g, s.items);
  return apertis_check_launch();
}
extern "C" int apertis_scan_lean_fwd(const float *dlt, const float *A_log, const void *Bt, int64_t bt_rs, const void *C, int64_t c_rs,
                                     const void *xc, int64_t xc_rs, const void *z, int64_t z_rs, const float *D, const float *h0,
                                     void *out, int64_t out_rs, float *h_last, float *agg, float *h_in, float *ckpt, int64_t B,
                                     int64_t L, int64_t h, int64_t N, int delta_softplus, void *stream) {
  return lean_fwd_impl(dlt, nullptr, 0, nullptr, nullptr, 0, A_log, Bt, bt_rs, C, c_rs, xc, xc_rs, z, z_rs, D, h0, out, out_rs, h_last,
                       agg, h_in, ckpt, B, L, h, N, delta_softplus, stream);
}
extern "C" int apertis_scan_lean_fwd_dt(const void *dt_in, int64_t dt_rs, const float *W_dt, const float *b_dt, int64_t R, float *dlt,
                                        const float *A_log, const void *Bt, int64_t bt_rs, const void *C, int64_t c_rs,
                                        const void *xc, int64_t xc_rs, const void *z, int64_t z_rs, const float *D, const float *h0,
                                        void *out, int64_t out_rs, float *h_last, float *agg, float *h_in, float *ckpt, int64_t B,
                                        int64_t L, int64_t h, int64_t N, int delta_softplus, void *stream) {
  if (!dt_in) return APERTIS_ERR_ARG;
  return lean_fwd_impl(dlt, dt_in, dt_rs, W_dt, b_dt, R, A_log, Bt, bt_rs, C, c_rs, xc, xc_rs, z, z_rs, D, h0, out, out_rs, h_last,
                       agg, h_in, ckpt, B, L, h, N, delta_softplus, stream);
}

extern "C" int apertis_scan_lean_bwd(const float *dlt, const float *A_log, const void *Bt, int64_t bt_rs, const void *C, int64_t c_rs,
                                     const void *xc, int64_t xc_rs, const void *z, int64_t z_rs, const float *D, const void *dout,
                                     int64_t dout_rs, const float *ckpt, void *dBt, int64_t dbt_rs, void *dC, int64_t dc_rs,
                                     int64_t store_w, void *dxc, int64_t dxc_rs, void *dz, int64_t dz_rs, float *d_dlt, float *dA_dD,
                                     float *agg, float *mu_in, float *fold, float *part, int64_t B, int64_t L, int64_t h, int64_t N,
                                     int delta_softplus, void *stream) {
  if (!dlt || !A_log || !Bt || !C || !xc || !z || !D || !dout || !ckpt || !dBt || !dC || !dxc || !dz || !d_dlt || !dA_dD || !agg ||
      !mu_in || !fold || !part)
    return APERTIS_ERR_ARG;
  LeanShape s;
  int rc = lean_shape(s, B, L, h, N, delta_softplus,
                      {{Bt, bt_rs}, {C, c_rs}, {xc, xc_rs}, {z, z_rs}, {dout, dout_rs}, {dxc, dxc_rs}, {dz, dz_rs}, {dBt, dbt_rs}, {dC, dc_rs}},
                      store_w);
  if (rc) return rc;
  const int64_t Dn = s.d.Dn, T = s.T;
  if (store_w < Dn || store_w > ceil_div64(Dn, TC) * TC || store_w % 4 || dbt_rs < store_w || dc_rs < store_w) return APERTIS_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  auto lt = [&](const void *p, int64_t rs, int64_t w) { return LeanT{p, (uint32_t)rs, (uint32_t)(((T - 1) * rs + w) * 2)}; };
  const LeanT tdl{dlt, (uint32_t)h, (uint32_t)(T * h * 4)};
  const unsigned grid = (unsigned)s.items;
  hipLaunchKernelGGL(scan_lean_bstate_k<LEAN_NW>, dim3(grid), dim3(64 * LEAN_NW), 0, st, tdl, A_log, lt(C, c_rs, Dn), lt(dout, dout_rs, Dn),
                     lt(z, z_rs, Dn), (float2 *)agg, s.d, s.g, s.items);
  hipLaunchKernelGGL(scan_lean_prefix_k<true>, dim3((unsigned)(B * s.g)), dim3(64), 0, st, (const float2 *)agg, (const float *)nullptr,
                     mu_in, s.d, s.g);
  hipLaunchKernelGGL(scan_lean_bwd_k, dim3(grid), dim3(64), 0, st, tdl, A_log, lt(Bt, bt_rs, Dn), lt(C, c_rs, Dn), lt(xc, xc_rs, Dn),
                     lt(z, z_rs, Dn), lt(dout, dout_rs, Dn), D, ckpt, mu_in, lt(dBt, dbt_rs, store_w), lt(dC, dc_rs, store_w), (int)store_w,
                     lt(dxc, dxc_rs, Dn), lt(dz, dz_rs, Dn), d_dlt, part, s.d, s.g, s.items);
  // fold the per-chunk partials [rows][2*Dn] (dA_log | dD) in a fixed order, two levels (as apertis_scan_gate_bwd does)
  const int64_t rows = s.d.B * s.d.nchunks, cols = 2 * Dn;
  const unsigned ctl = (unsigned)ceil_div64(cols, TC);
  if (rows <= 128) {
    hipLaunchKernelGGL(colsum_kernel, dim3(ctl), dim3(1024), 0, st, part, dA_dD, rows, cols, rows);
  } else {
    const int64_t groups = std::min<int64_t>(64, ceil_div64(rows, 64)), rpg = ceil_div64(rows, groups);
    const int64_t ng = ceil_div64(rows, rpg);
    hipLaunchKernelGGL(colsum_kernel, dim3(ctl, (unsigned)ng), dim3(1024), 0, st, part, fold, rows, cols, rpg);
    hipLaunchKernelGGL(colsum_kernel, dim3(ctl), dim3(1024), 0, st, fold, dA_dD, ng, cols, ng);
  }
  return apertis_check_launch();
}

extern "C" int apertis_ssm_decode_conv(const void *xp, int64_t xp_rs, const void *conv_state, void *conv_state_out,
                                       const float *w, const float *bias, void *xc, int64_t B, int64_t Dn, int64_t k,
                                       int dtype, void *stream) {
  if (!xp || !conv_state_out || !w || !bias || !xc || B <= 0 || Dn <= 0 || k < 1 || k > 16 || xp_rs < Dn) return APERTIS_ERR_ARG;
  if (k > 1 && !conv_state) return APERTIS_ERR_ARG;
  const unsigned grid = (unsigned)ceil_div64(B * Dn, 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == APERTIS_F32)
    hipLaunchKernelGGL(decode_conv_k<float>, dim3(grid), dim3(256), 0, st, (const float *)xp, xp_rs, (const float *)conv_state,
                       (float *)conv_state_out, w, bias, (float *)xc, B, Dn, (int)k);
  else if (dtype == APERTIS_BF16)
    hipLaunchKernelGGL(decode_conv_k<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t *)xp, xp_rs, (const bf16_t *)conv_state,
                       (bf16_t *)conv_state_out, w, bias, (bf16_t *)xc, B, Dn, (int)k);
  else
    return APERTIS_ERR_UNSUPPORTED;
  return apertis_check_launch();
}

static int decode_state_impl(const float *dt_logits, const void *dt_in, int64_t dt_rs, const float *W_dt, const float *b_dt, int64_t R,
                             const float *A_log, const void *Bt, int64_t bt_rs,
                             const void *C, int64_t c_rs, const void *xc, const void *z, int64_t z_rs,
                             const float *D, float *state, void *out, int64_t B, int64_t h, int64_t N, int dtype,
                             int delta_softplus, void *stream) {
  if ((!dt_logits && !dt_in) || !A_log || !Bt || !C || !xc || !z || !D || !state || !out || B <= 0 || h <= 0 || N <= 0) return APERTIS_ERR_ARG;
  if (dt_in && (!W_dt || R < 1 || R > 4096 || dt_rs < R)) return APERTIS_ERR_ARG;
  const int64_t Dn = h * N;
  if (bt_rs < Dn || c_rs < Dn || z_rs < Dn) return APERTIS_ERR_ARG;
  const unsigned grid = (unsigned)ceil_div64(B * Dn, 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == APERTIS_F32)
    hipLaunchKernelGGL(decode_state_k<float>, dim3(grid), dim3(256), 0, st, dt_logits, A_log, (const float *)Bt, bt_rs,
                       (const float *)C, c_rs, (const float *)xc, (const float *)z, z_rs, D, state, (float *)out, B, h, N,
                       delta_softplus, (const float *)dt_in, dt_rs, W_dt, b_dt, (int)R);
  else if (dtype == APERTIS_BF16)
    hipLaunchKernelGGL(decode_state_k<bf16_t>, dim3(grid), dim3(256), 0, st, dt_logits, A_log, (const bf16_t *)Bt, bt_rs,
                       (const bf16_t *)C, c_rs, (const bf16_t *)xc, (const bf16_t *)z, z_rs, D, state, (bf16_t *)out, B, h, N,
                       delta_softplus, (const bf16_t *)dt_in, dt_rs, W_dt, b_dt, (int)R);
  else
    return APERTIS_ERR_UNSUPPORTED;
  return apertis_check_launch();
}
extern "C" int apertis_ssm_decode_state(const float *dt_logits, const float *A_log, const void *Bt, int64_t bt_rs,
                                        const void *C, int64_t c_rs, const void *xc, const void *z, int64_t z_rs,
                                        const float *D, float *state, void *out, int64_t B, int64_t h, int64_t N, int dtype,
                                        int delta_softplus, void *stream) {
  if (!dt_logits) return APERTIS_ERR_ARG;
  return decode_state_impl(dt_logits, nullptr, 0, nullptr, nullptr, 0, A_log, Bt, bt_rs, C, c_rs, xc, z, z_rs, D, state, out, B, h, N,
                           dtype, delta_softplus, stream);
}
extern "C" int apertis_ssm_decode_state_dt(const void *dt_in, int64_t dt_rs, const float *W_dt, const float *b_dt, int64_t R,
                                           const float *A_log, const void *Bt, int64_t bt_rs, const void *C, int64_t c_rs,
                                           const void *xc, const void *z, int64_t z_rs, const float *D, float *state, void *out,
                                           int64_t B, int64_t h, int64_t N, int dtype, int delta_softplus, void *stream) {
  if (!dt_in) return APERTIS_ERR_ARG;
  return decode_state_impl(nullptr, dt_in, dt_rs, W_dt, b_dt, R, A_log, Bt, bt_rs, C, c_rs, xc, z, z_rs, D, state, out, B, h, N,
                           dtype, delta_softplus, stream);
}
