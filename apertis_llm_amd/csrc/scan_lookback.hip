// Selective scan + skip + gate, ONE launch per direction with a decoupled look-back (round 5; bf16, N = 16, Dn <= 256).
//
// Reference: SelectiveLinearAttention.forward, /root/reference/src/model/core.py:337-353 (the recurrence) and :394-397
// (skip + gate); the backward is the adjoint of that loop (SURVEY.md 8(a) row S-bwd).
//
// The lean kernels of scan_gate.hip (round 4) run three launches per direction: a state pass that reads Bt + delta (resp. C,
// dout, z) only to form the chunk aggregates, a prefix launch, and the replay that reads the same rows AGAIN - 1.53x / 1.59x the
// algorithmic bytes on the memory-side counters, which is what held them at 45 % / 40 % of the HBM peak while they streamed
// 5.0-5.5 TB/s of counter bytes.  Here every operand row is read ONCE:
//   * a work-group owns one 64-token chunk of a sequence, each of its 4 waves 16 consecutive tokens; a lane owns four
//     consecutive channels of a row as in the lean kernels (whole rows per wave-instruction, no LDS staging, no transpose);
//     models with Dn <= 128 put 64 / (Dn / 4) sequences side by side in a wave;
//   * a wave loads ALL its rows into registers up front (the replay re-uses the copy the aggregate was formed from), forms the
//     aggregate (sum of delta, state from zero) of its 16 tokens, and the four meet in LDS;
//   * the chunk's aggregate is PUBLISHED as 8-byte {epoch, value} granules (cdna_hip_programming.md Guideline 16, form R2:
//     the data is the flag), two per 16-byte write-through store; the last chunk of every super-chunk of 4 also publishes the
//     INCLUSIVE state at the super-chunk's end.  The carry entering a chunk = that inclusive record of the super-chunk in
//     front + the aggregates of the (<= 3) chunks of its own super-chunk before it: four records, polled by the four waves
//     in parallel (wave 0 the inclusive one, waves 1-3 a sibling each), handed round through LDS;
//   * work-groups take their chunk from a ticket counter in chunk-major order and only ever wait for records of LOWER
//     tickets, whose owners have started and never wait for a higher one: progress does not depend on residency or dispatch
//     order.  The composition order is fixed (tokens, waves, siblings, super-chunks: always left to right resp. right to
//     left), so the result is run-to-run identical.  Waits are bounded; a time-out sets the workspace's error word.
// Extra traffic per 64-token chunk (115 KB of operands forward at Dn = 176): one 2.1 KB aggregate record written, <= 3 read,
// 1.4 KB inclusive record per four chunks written and one read: ~5 %.  The forward leaves the state entering every 16th
// token (44 B per token at Dn = 176, 2.4 % of its bytes: `ckpt16`) for the backward, which rebuilds a wave's 16 states from
// it instead of reading a checkpoint every fourth token (round 4: 9.7 %).
//
// Algorithmic bytes per token (fused epilogue variant): forward 5*Dn*2 + 4h, backward 9*Dn*2 + 8h.
#include "scan_lean.h"

namespace {

constexpr int LB_LT = LT_DEFAULT;       // tokens per chunk (= apertis_scan_gate_chunk_len(): the granularity of h_in)
constexpr int LB_NW = 4;                // waves per chunk
constexpr int LB_TW = LB_LT / LB_NW;    // tokens per wave
constexpr int LB_SUP = 4;               // chunks per super-chunk = records a chunk gathers = waves that poll
static_assert(LB_SUP == LB_NW && LB_TW == 16 && LB_LT == 64, "the polling roles and the unrolled token loops assume 4 x 16 tokens");
constexpr unsigned LB_SPIN_MAX = 1u << 16;    // (~0.1 s of polling: a legitimate wait is under a millisecond)
#ifndef LB_LATE_
#define LB_LATE_ 8
#endif
#ifndef LB_BWD_LAUNDER_DECAY
#define LB_BWD_LAUNDER_DECAY 1
#endif
#ifndef LB_SCHED_TOKEN
#define LB_SCHED_TOKEN 1
#endif
constexpr int LB_LATE = LB_LATE_;
static_assert(LB_LATE % 4 == 0 && LB_LATE >= 0 && LB_LATE <= 12, "whole groups of four tokens");        // forward: tokens whose C, xc, z loads are issued inside the replay

typedef unsigned lb_u4 __attribute__((ext_vector_type(4)));

// geometry of a launch: g lanes per row (Dn / 4), R rows (sequences) side by side in a wave, nbw waves' worth of sequences,
// nsup super-chunks, nck16 checkpoint rows per sequence; byte offsets of the two record tables in the workspace
struct LbGeo { int g, R, nbw, nsup, nck16; uint32_t offA, offI, wsbytes; };

// records: aggregate of chunk k of sequence b = three 16-byte pieces per lane {e, S0, e, S1} {e, S2, e, S3} {e, sd, e, sd}
// at offA + (((b * nchunks + k) * 3 + piece) * g + q) * 16; inclusive state after super-chunk s = the first two pieces at
// offI + (((b * nsup + s) * 2 + piece) * g + q) * 16.  (k, s count in composition order: right to left in the backward.)
__device__ __forceinline__ void lb_store16(__amdgpu_buffer_rsrc_t rs, uint32_t off, uint32_t e, float a, float b) {
  const lb_u4 v = {e, __float_as_uint(a), e, __float_as_uint(b)};
  __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off, 0, 16);     // aux 16 = sc1: write-through
}
// The poll's loads are inline asm: hipcc hoists __builtin_amdgcn_raw_buffer_load_b128 out of the poll loop (with an empty
// memory-clobber asm in the loop, and with the builtin's volatile bit as well) and the loop then spins on registers.  The
// descriptor as four plain words for the "s" operand; the wait names the loaded registers so that their uses stay behind it.
typedef int lb_i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ lb_i4 lb_desc(const void *p, uint32_t bytes) {
  const uint64_t a = (uint64_t)(uintptr_t)p;
  const lb_i4 r = {(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
  return r;
}
__device__ __forceinline__ lb_u4 lb_load16(lb_i4 rs, uint32_t off) {
  lb_u4 v;
  // (s_nop 4: hipcc's hazard recogniser does not look inside inline asm - a descriptor word restored from a spill lane by
  //  v_readlane right in front of this would be read by the load before it has landed: five wait states, gfx9 VALU-SGPR -> VMEM)
  asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen sc1" : "=v"(v) : "v"(off), "s"(rs) : "memory");   // sc1: past this CU's L1
  return v;
}
// LDS-only barrier: the waves' row loads (and output stores) stay in flight across it
__device__ __forceinline__ void lb_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct LbLane { bool ok; int b, q, c0, hh; };


// ticket -> (position in composition order, wave of sequences); lanes -> (sequence, four channels)
__device__ __forceinline__ bool lb_take(GateWsHead *head, uint32_t epoch, int *s_item, const ScanDims &d, const LbGeo &G, int &k,
                                        LbLane &ln_) {
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(&head->ctr[epoch & 1], 1u);
    if (t == 0) atomicExch(&head->ctr[(epoch + 1) & 1], 0u);   // the next launch's counter (its last user has finished)
    *s_item = (int)t;
  }
  __syncthreads();
  const int item = __builtin_amdgcn_readfirstlane(*s_item);     // (wave-uniform: the rows' token terms ride in scalar offsets)
  const int ln = (int)threadIdx.x & 63;
  if (item >= d.nchunks * G.nbw) return false;
  k = item / G.nbw;
  const int bw = item - k * G.nbw;
  const int r = ln / G.g;
  ln_.q = ln - r * G.g;
  ln_.b = bw * G.R + r;
  ln_.ok = r < G.R && ln_.b < d.B;
  ln_.c0 = 4 * ln_.q;
  ln_.hh = ln_.c0 >> 4;
  return true;
}

// one record polled by one wave: pieces [0, np) at `off` (+ piece * g * 16), until every granule carries this launch's epoch
template <int NP>
__device__ __forceinline__ void lb_poll(lb_i4 rw, uint32_t off, uint32_t pstep, uint32_t epoch, bool lane_ok,
                                        int *err, float (&S)[4], float &sd) {
  static_assert(NP == 2 || NP == 3, "an inclusive record or an aggregate");
  lb_u4 p[NP];
  unsigned spins = 0;
  while (true) {
#pragma unroll
    for (int i = 0; i < NP; ++i) p[i] = lb_load16(rw, lane_ok ? off + (uint32_t)i * pstep : 0xfffffff0u);   // (select LAST: bad + step wraps)
    if constexpr (NP == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(p[0]), "+v"(p[1]) : : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]) : : "memory");
    bool okk = true;
#pragma unroll
    for (int i = 0; i < NP; ++i) okk = okk && p[i][0] == epoch && p[i][2] == epoch;
    if (__all(okk || !lane_ok)) break;
    if (++spins > LB_SPIN_MAX) { if ((threadIdx.x & 63) == 0) atomicOr(err, 2); break; }
    __builtin_amdgcn_s_sleep(2);
  }
  S[0] = __uint_as_float(p[0][1]); S[1] = __uint_as_float(p[0][3]); S[2] = __uint_as_float(p[1][1]); S[3] = __uint_as_float(p[1][3]);
  sd = NP > 2 ? __uint_as_float(p[NP - 1][1]) : 0.f;
}


// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
#ifndef LB_FWD_MINW
#define LB_FWD_MINW 3
#endif
__global__ void __launch_bounds__(64 * LB_NW, LB_FWD_MINW)
scan_lb_fwd_k(LeanT dl, const float *__restrict__ A_log, LeanT tb_, LeanT tc, LeanT tx, LeanT tz, const float *__restrict__ Dv,
              const float *__restrict__ h0, float *__restrict__ h_in, float *__restrict__ h_last, float *__restrict__ ckpt, LeanT to,
              GateWsHead *__restrict__ head, uint32_t epoch, ScanDims d, LbGeo G) {
  __shared__ float4 sS[2][LB_NW][64];    // [0]: the waves' aggregates (state from zero); [1]: the records the waves polled
  __shared__ float sD[2][LB_NW][64];     // ... and their sums of delta
  __shared__ int s_item;
  int chunk;
  LbLane L;
  if (!lb_take(head, epoch, &s_item, d, G, chunk, L)) return;
  const int ln = (int)threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), qi = ln & 3;
  const int sup = chunk / LB_SUP, sib = chunk - sup * LB_SUP;
  const int t0 = chunk * LB_LT + wv * LB_TW;
  const int rows = min(LB_TW, (int)d.L - t0);              // (<= 0: this wave's tokens are all past the end)
  const bool ragged = rows < LB_TW;
  const int tlast = max(rows - 1, 0);
  const uint32_t bad = 0xfffffff0u;                         // an offset no descriptor covers: the access is dropped (loads: zeros)
  float A2[4] = {0.f, 0.f, 0.f, 0.f}, Dk[4] = {0.f, 0.f, 0.f, 0.f};
  if (L.ok) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { A2[k] = -expf(A_log[L.c0 + k]) * LOG2E_F; Dk[k] = Dv[L.c0 + k]; }
  }
  const __amdgpu_buffer_rsrc_t rb = lean_rsrc(tb_.p, tb_.bytes), rc = lean_rsrc(tc.p, tc.bytes), rx = lean_rsrc(tx.p, tx.bytes),
                               rz = lean_rsrc(tz.p, tz.bytes), rd = lean_rsrc(dl.p, dl.bytes), ro = lean_rsrc(to.p, to.bytes),
                               rw = lean_rsrc(head, G.wsbytes);
  const lb_i4 rwp = lb_desc(head, G.wsbytes);
  const uint32_t tok0 = (uint32_t)((int64_t)L.b * d.L + t0), c2 = (uint32_t)L.c0 * 2u;
  const uint32_t ob = L.ok ? tok0 * tb_.rs * 2u + c2 : bad, oc = L.ok ? tok0 * tc.rs * 2u + c2 : bad,
                 ox = L.ok ? tok0 * tx.rs * 2u + c2 : bad, oz = L.ok ? tok0 * tz.rs * 2u + c2 : bad,
                 oo = L.ok ? tok0 * to.rs * 2u + c2 : bad;
  // ---- every row of this wave's 16 tokens, issued at once (delta and Bt first: the aggregate needs only them).  The token's
  // row term rides in the scalar offset (the hardware range check covers the lane offset only, so a token past the sequence's
  // end re-reads the last row); such a token gets delta = 0 (a = 1) and its Bt registers are zeroed where the aggregate pass
  // first meets them: the state passes through it and nothing else in the token loops needs a mask ----
  float vd[4];
  uint2 vb[LB_TW], vc[LB_TW], vx[LB_TW], vz[LB_TW];
#pragma unroll
  for (int j = 0; j < 4; ++j)    // lane qi of a quad takes token 4j + qi (the token rides in the lane offset here)
    vd[j] = lean_ld4(rd, L.ok ? ((tok0 + (uint32_t)min(4 * j + qi, tlast)) * dl.rs + (uint32_t)L.hh) * 4u : bad, 0u);
#pragma unroll
  for (int u = 0; u < LB_TW; ++u) vb[u] = lean_ld8(rb, ob, (uint32_t)min(u, tlast) * tb_.rs * 2u);
  auto ld_late = [&](int u) {
    const uint32_t t = (uint32_t)min(u, tlast);
    vc[u] = lean_ld8(rc, oc, t * tc.rs * 2u);
    vx[u] = lean_ld8(rx, ox, t * tx.rs * 2u);
    vz[u] = lean_ld8(rz, oz, t * tz.rs * 2u);
  };
  // (the last LB_LATE tokens' C, xc, z follow group by group inside the replay, into the registers it frees: everything at once
  //  is 173 VGPRs against the 168 of three waves per SIMD)
#pragma unroll
  for (int u = 0; u < LB_TW - LB_LATE; ++u) ld_late(u);
  // ---- aggregate of the 16 tokens: (sum of delta, state from zero) ----
  float sp[4], S[4] = {0.f, 0.f, 0.f, 0.f}, sumdl = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sp[j] = d.softplus ? softplus_fast(vd[j]) : vd[j];
    if (ragged) sp[j] = 4 * j + qi < rows ? sp[j] : 0.f;   // a token past the end: delta 0 -> a = 1 (and its Bt reads as 0)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = 4 * j + i;
      const float dlv = quad_bc(sp[j], i);
      float bv[4];
      vb[u].x = u < rows ? vb[u].x : 0u; vb[u].y = u < rows ? vb[u].y : 0u;
      unpack4(vb[u], bv);
      sumdl += dlv;
#pragma unroll
      for (int k = 0; k < 4; ++k) S[k] = fmaf(__builtin_amdgcn_exp2f(dlv * A2[k]), S[k], bv[k]);
    }
  }
  sS[0][wv][ln] = make_float4(S[0], S[1], S[2], S[3]);
  sD[0][wv][ln] = sumdl;
  lb_barrier();
  // ---- wave 0 publishes the chunk's aggregate; every wave polls one record of the carry ----
  const uint32_t pstep = (uint32_t)G.g * 16u;
  float Sa[4] = {0.f, 0.f, 0.f, 0.f}, sda = 0.f;
  if (wv == 0) {
#pragma unroll
    for (int w = 0; w < LB_NW; ++w) {
      const float4 s4 = sS[0][w][ln];
      const float sw = sD[0][w][ln], sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) Sa[k] = fmaf(__builtin_amdgcn_exp2f(sw * A2[k]), Sa[k], sv[k]);
      sda += sw;
    }
    const uint32_t o = G.offA + ((uint32_t)(L.b * d.nchunks + chunk) * 3u * (uint32_t)G.g + (uint32_t)L.q) * 16u;
    lb_store16(rw, L.ok ? o : bad, epoch, Sa[0], Sa[1]);          // (select LAST: bad + pstep would wrap into the table)
    lb_store16(rw, L.ok ? o + pstep : bad, epoch, Sa[2], Sa[3]);
    lb_store16(rw, L.ok ? o + 2u * pstep : bad, epoch, sda, sda);
    float hv[4] = {0.f, 0.f, 0.f, 0.f}, unused;
    if (sup > 0) {
      lb_poll<2>(rwp, G.offI + ((uint32_t)(L.b * G.nsup + sup - 1) * 2u * (uint32_t)G.g + (uint32_t)L.q) * 16u, pstep, epoch,
                 L.ok, &head->err, hv, unused);
    } else if (h0 && L.ok) {
      const float4 t = *reinterpret_cast<const float4 *>(h0 + (int64_t)L.b * d.Dn + L.c0);
      hv[0] = t.x; hv[1] = t.y; hv[2] = t.z; hv[3] = t.w;
    }
    sS[1][0][ln] = make_float4(hv[0], hv[1], hv[2], hv[3]);
  } else if (wv - 1 < sib) {
    float rv[4], rsd;
    lb_poll<3>(rwp, G.offA + ((uint32_t)(L.b * d.nchunks + sup * LB_SUP + wv - 1) * 3u * (uint32_t)G.g + (uint32_t)L.q) * 16u,
               pstep, epoch, L.ok, &head->err, rv, rsd);
    sS[1][wv][ln] = make_float4(rv[0], rv[1], rv[2], rv[3]);
    sD[1][wv][ln] = rsd;
  }
  lb_barrier();
  // ---- the state entering the chunk, then this wave's tokens ----
  float hst[4];
  {
    const float4 t = sS[1][0][ln];
    hst[0] = t.x; hst[1] = t.y; hst[2] = t.z; hst[3] = t.w;
  }
#pragma unroll
  for (int i = 0; i < LB_SUP - 1; ++i)
    if (i < sib) {
      const float4 s4 = sS[1][i + 1][ln];
      const float sw = sD[1][i + 1][ln], sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) hst[k] = fmaf(__builtin_amdgcn_exp2f(sw * A2[k]), hst[k], sv[k]);
    }
  if (wv == 0) {
    if (sib == LB_SUP - 1) {      // the inclusive state after this super-chunk: the next one's chunks wait for it
      const uint32_t o = G.offI + ((uint32_t)(L.b * G.nsup + sup) * 2u * (uint32_t)G.g + (uint32_t)L.q) * 16u;
      float hi[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) hi[k] = fmaf(__builtin_amdgcn_exp2f(sda * A2[k]), hst[k], Sa[k]);
      lb_store16(rw, L.ok ? o : bad, epoch, hi[0], hi[1]);
      lb_store16(rw, L.ok ? o + pstep : bad, epoch, hi[2], hi[3]);
    }
    if (h_in && L.ok)
      *reinterpret_cast<float4 *>(h_in + ((int64_t)L.b * d.nchunks + chunk) * d.Dn + L.c0) = make_float4(hst[0], hst[1], hst[2], hst[3]);
  }
#pragma unroll
  for (int w = 0; w < LB_NW - 1; ++w)
    if (w < wv) {
      const float4 s4 = sS[0][w][ln];
      const float sw = sD[0][w][ln], sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) hst[k] = fmaf(__builtin_amdgcn_exp2f(sw * A2[k]), hst[k], sv[k]);
    }
  if (ckpt && L.ok && rows > 0)
    *reinterpret_cast<float4 *>(ckpt + ((int64_t)L.b * G.nck16 + (t0 >> 4)) * d.Dn + L.c0) = make_float4(hst[0], hst[1], hst[2], hst[3]);
  // ---- replay from the register copy, skip + gate ----
  // (the copy is laundered through an empty asm: hipcc otherwise keeps the aggregate pass's unpacked Bt and its 64 decay
  //  factors alive for this pass - common subexpressions - and needs 212 VGPRs)
#pragma unroll
  for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(sp[j]));
#pragma unroll
  for (int u = 0; u < LB_TW; ++u) asm volatile("" : "+v"(vb[u].x), "+v"(vb[u].y));
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = 4 * j + i;
      const float dlv = quad_bc(sp[j], i);
      float bv[4], cv[4], xv[4], zv[4], o[4];
      unpack4(vb[u], bv); unpack4(vc[u], cv); unpack4(vx[u], xv); unpack4(vz[u], zv);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float a = __builtin_amdgcn_exp2f(dlv * A2[k]);
        hst[k] = fmaf(a, hst[k], bv[k]);
        const float y = cv[k] * hst[k];
        const float dx = Dk[k] * xv[k];
        const float val = y + dx;
        o[k] = val * silu_g(zv[k]);
      }
      const lean_u2 ov = {lean_pack2(o[0], o[1]), lean_pack2(o[2], o[3])};
      if (u < rows) __builtin_amdgcn_raw_buffer_store_b64(ov, ro, (int)oo, (int)((uint32_t)u * to.rs * 2u), 2);   // (wave-uniform branch)
#if LB_SCHED_TOKEN
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    if (j + (LB_TW - LB_LATE) / 4 < 4) {     // the group (LB_TW - LB_LATE) / 4 groups ahead, into the registers this one frees
#pragma unroll
      for (int u = 4 * j + LB_TW - LB_LATE; u < 4 * j + LB_TW - LB_LATE + 4; ++u) ld_late(u);
    }
    __builtin_amdgcn_sched_barrier(0);   // (four tokens at a time: hipcc otherwise unpacks and gates many tokens ahead of the state chain and spills)
  }
  if (h_last && wv == LB_NW - 1 && chunk == d.nchunks - 1 && L.ok)
    *reinterpret_cast<float4 *>(h_last + (int64_t)L.b * d.Dn + L.c0) = make_float4(hst[0], hst[1], hst[2], hst[3]);
}

// ---------------------------------------------------------------------------------------------------------------
// backward: chunks right to left (position k = nchunks - 1 - chunk), waves right to left inside a chunk (rank wr = 3 - wave).
//   aggregate of 16 tokens: (sum of delta, M) with M = mu at the first token from zero entering at the right end,
//       mu_t = a_t lambda_t,  lambda_t = dv_t C_t + mu_{t+1},  dv = dout silu(z)
//   replay: the states are rebuilt from the state entering the wave's first token (the forward's ckpt16) - one pass over Bt for
//   the entry states of the four blocks of four tokens, then each block's four states right before its adjoint walk (a second
//   pass over Bt instead of 48 more registers) -, and the adjoint walk emits
//   dBt = lambda, dC = dv s, dxc = dv D, dz = dout silu'(z) (C s + D xc), d delta = sum over the head of lambda s_{t-1} a A (through
//   the softplus), and the chunk's partial sums of dA_log and dD (folded by colsum_kernel in a fixed order).
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64 * LB_NW, 2)
scan_lb_bwd_k(LeanT dl, const float *__restrict__ A_log, LeanT tb_, LeanT tc, LeanT tx, LeanT tz, LeanT tg, const float *__restrict__ Dv,
              const float *__restrict__ ckpt, LeanT ob_, LeanT oc_, int store_w, LeanT ox_, LeanT oz_, float *__restrict__ d_dlt,
              float *__restrict__ part, GateWsHead *__restrict__ head, uint32_t epoch, ScanDims d, LbGeo G) {
  __shared__ float4 sS[2][LB_NW][64];
  __shared__ float sD[2][LB_NW][64];
  __shared__ float4 sP[2][LB_NW][64];    // the waves' partial sums of dA_log, dD
  __shared__ int s_item;
  int kpos;
  LbLane L;
  if (!lb_take(head, epoch, &s_item, d, G, kpos, L)) return;
  const int chunk = d.nchunks - 1 - kpos;
  const int ln = (int)threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), wr = LB_NW - 1 - wv, qi = ln & 3;
  const int sup = kpos / LB_SUP, sib = kpos - sup * LB_SUP;
  const int t0 = chunk * LB_LT + wv * LB_TW;
  const int rows = min(LB_TW, (int)d.L - t0);
  const bool ragged = rows < LB_TW;
  const int tlast = max(rows - 1, 0);
  const uint32_t bad = 0xfffffff0u;
  // (A2 = A log2(e) serves the decay exp2(delta A2) AND, with a factor ln 2 applied once per sum, the A of da_t a_t A: no second
  //  copy of A in registers)
  float A2[4] = {0.f, 0.f, 0.f, 0.f}, Dk[4] = {0.f, 0.f, 0.f, 0.f};
  if (L.ok) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { A2[k] = -expf(A_log[L.c0 + k]) * LOG2E_F; Dk[k] = Dv[L.c0 + k]; }
  }
  constexpr float LN2_F = 0.6931471805599453f;
  const __amdgpu_buffer_rsrc_t rb = lean_rsrc(tb_.p, tb_.bytes), rc = lean_rsrc(tc.p, tc.bytes), rx = lean_rsrc(tx.p, tx.bytes),
                               rz = lean_rsrc(tz.p, tz.bytes), rg = lean_rsrc(tg.p, tg.bytes), rd = lean_rsrc(dl.p, dl.bytes),
                               rw = lean_rsrc(head, G.wsbytes);
  const lb_i4 rwp = lb_desc(head, G.wsbytes);
  const __amdgpu_buffer_rsrc_t wb = lean_rsrc(ob_.p, ob_.bytes), wc = lean_rsrc(oc_.p, oc_.bytes), wx = lean_rsrc(ox_.p, ox_.bytes),
                               wz = lean_rsrc(oz_.p, oz_.bytes);
  const uint32_t tok0 = (uint32_t)((int64_t)L.b * d.L + t0), c2 = (uint32_t)L.c0 * 2u;
  const uint32_t fb = L.ok ? tok0 * tb_.rs * 2u + c2 : bad, fc = L.ok ? tok0 * tc.rs * 2u + c2 : bad, fx = L.ok ? tok0 * tx.rs * 2u + c2 : bad,
                 fz = L.ok ? tok0 * tz.rs * 2u + c2 : bad, fg = L.ok ? tok0 * tg.rs * 2u + c2 : bad;
  // (store offsets with the lane mask folded in once; a token past the sequence's end skips its stores by a wave-uniform branch -
  //  a select per token and tensor was hoisted by hipcc into 96 live registers)
  const uint32_t sb = L.ok ? tok0 * ob_.rs * 2u + c2 : bad, sc = L.ok ? tok0 * oc_.rs * 2u + c2 : bad,
                 sx = L.ok ? tok0 * ox_.rs * 2u + c2 : bad, sz = L.ok ? tok0 * oz_.rs * 2u + c2 : bad;
  // ---- the rows of this wave's 16 tokens (delta, C, dout, z first: the aggregate needs only them).  A token past the
  // sequence's end re-reads the last row, gets delta = 0 (a = 1) and its dout registers are zeroed where the aggregate pass first
  // meets them: the adjoint passes through it and nothing else in the token loops needs a mask (its stores are dropped) ----
  float vd[4];
  uint2 vc[LB_TW], vg[LB_TW], vz[LB_TW], vb[LB_TW], vx[LB_TW];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    vd[j] = lean_ld4(rd, L.ok ? ((tok0 + (uint32_t)min(4 * j + qi, tlast)) * dl.rs + (uint32_t)L.hh) * 4u : bad, 0u);
#pragma unroll
  for (int u = LB_TW - 1; u >= 0; --u) {         // (right to left: the aggregate pass walks that way)
    const uint32_t t = (uint32_t)min(u, tlast);
    vc[u] = lean_ld8(rc, fc, t * tc.rs * 2u);
    vg[u] = lean_ld8(rg, fg, t * tg.rs * 2u);
    vz[u] = lean_ld8(rz, fz, t * tz.rs * 2u);
  }
  float4 vk = make_float4(0.f, 0.f, 0.f, 0.f);
  if (L.ok && rows > 0) vk = *reinterpret_cast<const float4 *>(ckpt + ((int64_t)L.b * G.nck16 + (t0 >> 4)) * d.Dn + L.c0);
#pragma unroll
  for (int u = 0; u < LB_TW; ++u) vb[u] = lean_ld8(rb, fb, (uint32_t)min(u, tlast) * tb_.rs * 2u);
  // (xc is only met in the adjoint walk, block of four tokens by block: the rightmost block's rows are issued behind the aggregate
  //  pass, every other block's under the walk of the block to its right - with all 80 rows in flight at once the kernel does not
  //  fit 256 VGPRs)
  auto ld_xc = [&](int blk) {
#pragma unroll
    for (int u = 4 * blk; u < 4 * blk + 4; ++u) vx[u] = lean_ld8(rx, fx, (uint32_t)min(u, tlast) * tx.rs * 2u);
  };
  // ---- reverse aggregate of the 16 tokens ----
  float sp[4], M[4] = {0.f, 0.f, 0.f, 0.f}, sumdl = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sp[j] = d.softplus ? softplus_fast(vd[j]) : vd[j];
    if (ragged) sp[j] = 4 * j + qi < rows ? sp[j] : 0.f;
  }
#pragma unroll
  for (int j = 3; j >= 0; --j) {
#pragma unroll
    for (int i = 3; i >= 0; --i) {
      const int u = 4 * j + i;
      const float dlv = quad_bc(sp[j], i);
      float cv[4], gv[4], zv[4];
      vg[u].x = u < rows ? vg[u].x : 0u; vg[u].y = u < rows ? vg[u].y : 0u;
      unpack4(vc[u], cv); unpack4(vg[u], gv); unpack4(vz[u], zv);
      sumdl += dlv;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float av = __builtin_amdgcn_exp2f(dlv * A2[k]);
        const float uu = (gv[k] * silu_g(zv[k])) * cv[k];
        M[k] = av * (uu + M[k]);
      }
      __builtin_amdgcn_sched_barrier(0);   // (a token at a time: hipcc otherwise gates many tokens ahead of the chain: +75 VGPRs)
    }
  }
  ld_xc(3);
  sS[0][wr][ln] = make_float4(M[0], M[1], M[2], M[3]);
  sD[0][wr][ln] = sumdl;
  lb_barrier();
  // ---- rank 0 (the rightmost wave) publishes the chunk's aggregate; every wave polls one record of the carry ----
  const uint32_t pstep = (uint32_t)G.g * 16u;
  float Ma[4] = {0.f, 0.f, 0.f, 0.f}, sda = 0.f;
  if (wr == 0) {
#pragma unroll
    for (int w = 0; w < LB_NW; ++w) {
      const float4 s4 = sS[0][w][ln];
      const float sw = sD[0][w][ln], sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) Ma[k] = fmaf(__builtin_amdgcn_exp2f(sw * A2[k]), Ma[k], sv[k]);
      sda += sw;
    }
    const uint32_t o = G.offA + ((uint32_t)(L.b * d.nchunks + kpos) * 3u * (uint32_t)G.g + (uint32_t)L.q) * 16u;
    lb_store16(rw, L.ok ? o : bad, epoch, Ma[0], Ma[1]);
    lb_store16(rw, L.ok ? o + pstep : bad, epoch, Ma[2], Ma[3]);
    lb_store16(rw, L.ok ? o + 2u * pstep : bad, epoch, sda, sda);
    float mv[4] = {0.f, 0.f, 0.f, 0.f}, unused;
    if (sup > 0)
      lb_poll<2>(rwp, G.offI + ((uint32_t)(L.b * G.nsup + sup - 1) * 2u * (uint32_t)G.g + (uint32_t)L.q) * 16u, pstep, epoch,
                 L.ok, &head->err, mv, unused);
    sS[1][0][ln] = make_float4(mv[0], mv[1], mv[2], mv[3]);
  } else if (wr - 1 < sib) {
    float rv[4], rsd;
    lb_poll<3>(rwp, G.offA + ((uint32_t)(L.b * d.nchunks + sup * LB_SUP + wr - 1) * 3u * (uint32_t)G.g + (uint32_t)L.q) * 16u,
               pstep, epoch, L.ok, &head->err, rv, rsd);
    sS[1][wr][ln] = make_float4(rv[0], rv[1], rv[2], rv[3]);
    sD[1][wr][ln] = rsd;
  }
  lb_barrier();
  float mu[4];
  {
    const float4 t = sS[1][0][ln];
    mu[0] = t.x; mu[1] = t.y; mu[2] = t.z; mu[3] = t.w;
  }
#pragma unroll
  for (int i = 0; i < LB_SUP - 1; ++i)
    if (i < sib) {
      const float4 s4 = sS[1][i + 1][ln];
      const float sw = sD[1][i + 1][ln], sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) mu[k] = fmaf(__builtin_amdgcn_exp2f(sw * A2[k]), mu[k], sv[k]);
    }
  if (wr == 0 && sib == LB_SUP - 1) {
    const uint32_t o = G.offI + ((uint32_t)(L.b * G.nsup + sup) * 2u * (uint32_t)G.g + (uint32_t)L.q) * 16u;
    float mi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) mi[k] = fmaf(__builtin_amdgcn_exp2f(sda * A2[k]), mu[k], Ma[k]);
    lb_store16(rw, L.ok ? o : bad, epoch, mi[0], mi[1]);
    lb_store16(rw, L.ok ? o + pstep : bad, epoch, mi[2], mi[3]);
  }
#pragma unroll
  for (int w = 0; w < LB_NW - 1; ++w)
    if (w < wr) {
      const float4 s4 = sS[0][w][ln];
      const float sw = sD[0][w][ln], sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) mu[k] = fmaf(__builtin_amdgcn_exp2f(sw * A2[k]), mu[k], sv[k]);
    }
  // ---- replay: block of four tokens by block, right to left: the block's four states rebuilt from its entry state, then the adjoint ----
  // (the register copy is laundered through an empty asm between the passes: hipcc otherwise keeps the aggregate pass's unpacked
  //  rows, gates and decay factors alive as common subexpressions and spills)
#pragma unroll
  for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(sp[j]));
#pragma unroll
  for (int u = 0; u < LB_TW; ++u)
    asm volatile("" : "+v"(vc[u].x), "+v"(vc[u].y), "+v"(vg[u].x), "+v"(vg[u].y), "+v"(vz[u].x), "+v"(vz[u].y));
  // the pad columns [Dn, store_w) of dBt / dC receive zeros (the padded slices of the projection output's gradient): the
  // first (store_w - Dn) / 4 lanes of the row write them
  const int npad = (store_w - (int)d.Dn) >> 2;
  const uint32_t pz = (uint32_t)(d.Dn + 4 * L.q) * 2u;
  const bool pad_lane = L.ok && L.q < npad;
  const uint32_t spb = tok0 * ob_.rs * 2u + pz, spc = tok0 * oc_.rs * 2u + pz;
  float *ddp = d_dlt + ((int64_t)tok0 + qi) * d.h + L.hh;
  const float hin[4] = {vk.x, vk.y, vk.z, vk.w};
  float dA[4] = {0.f, 0.f, 0.f, 0.f}, dD[4] = {0.f, 0.f, 0.f, 0.f};
  // entry states of the four blocks of four tokens: one pass over the Bt of tokens 0 .. 11
  float hb[4][4];
  {
    float hc[4] = {hin[0], hin[1], hin[2], hin[3]};
#pragma unroll
    for (int k = 0; k < 4; ++k) hb[0][k] = hin[k];
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      const float dlv = quad_bc(sp[u >> 2], u & 3);
      float bv[4];
      unpack4(vb[u], bv);
#pragma unroll
      for (int k = 0; k < 4; ++k) hc[k] = fmaf(__builtin_amdgcn_exp2f(dlv * A2[k]), hc[k], bv[k]);
      if ((u & 3) == 3) {
#pragma unroll
        for (int k = 0; k < 4; ++k) hb[(u >> 2) + 1][k] = hc[k];
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 12; ++u) asm volatile("" : "+v"(vb[u].x), "+v"(vb[u].y));
#pragma unroll
  for (int j = 3; j >= 0; --j) {          // tokens 4j .. 4j+3, right to left
    if (j > 0) ld_xc(j - 1);              // (the next block's xc under this block's walk)
    float hs[4][4];
    {
      float hc[4] = {hb[j][0], hb[j][1], hb[j][2], hb[j][3]};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int u = 4 * j + e;
        const float dlv = quad_bc(sp[j], e);
        float bv[4];
        unpack4(vb[u], bv);
#pragma unroll
        for (int k = 0; k < 4; ++k) { hc[k] = fmaf(__builtin_amdgcn_exp2f(dlv * A2[k]), hc[k], bv[k]); hs[e][k] = hc[k]; }
      }
    }
#if LB_BWD_LAUNDER_DECAY
    asm volatile("" : "+v"(sp[j]));   // (the adjoint recomputes its decay factors)
#endif
    float ddl_keep = 0.f;
#pragma unroll
    for (int i = 3; i >= 0; --i) {
      const int u = 4 * j + i;
      const float dlv = quad_bc(sp[j], i);
      float cv[4], xv[4], zv[4], gv[4], oB[4], oC[4], oX[4], oZ[4];
      unpack4(vc[u], cv); unpack4(vx[u], xv); unpack4(vz[u], zv); unpack4(vg[u], gv);
      float qs = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float av = __builtin_amdgcn_exp2f(dlv * A2[k]);
        float f, df;
        silu_both(zv[k], f, df);
        const float dv = gv[k] * f;
        const float lam = fmaf(dv, cv[k], mu[k]);
        const float hprev = i > 0 ? hs[i - 1][k] : hb[j][k];
        const float qq = lam * hprev * av * A2[k];                 // da_t * a_t * A (x log2 e: taken out below)
        const float y = cv[k] * hs[i][k];
        const float dx = Dk[k] * xv[k];
        const float val = y + dx;
        oB[k] = lam;
        oC[k] = dv * hs[i][k];
        oX[k] = dv * Dk[k];
        oZ[k] = gv[k] * df * val;
        dA[k] = fmaf(qq, dlv, dA[k]);
        dD[k] = fmaf(dv, xv[k], dD[k]);
        qs += qq;
        mu[k] = av * lam;
      }
      const uint32_t ts = (uint32_t)u;
      const lean_u2 wB = {lean_pack2(oB[0], oB[1]), lean_pack2(oB[2], oB[3])}, wC = {lean_pack2(oC[0], oC[1]), lean_pack2(oC[2], oC[3])},
                    wX = {lean_pack2(oX[0], oX[1]), lean_pack2(oX[2], oX[3])}, wZ = {lean_pack2(oZ[0], oZ[1]), lean_pack2(oZ[2], oZ[3])};
      if (u < rows) {
        __builtin_amdgcn_raw_buffer_store_b64(wB, wb, (int)sb, (int)(ts * ob_.rs * 2u), 2);
        __builtin_amdgcn_raw_buffer_store_b64(wC, wc, (int)sc, (int)(ts * oc_.rs * 2u), 2);
        __builtin_amdgcn_raw_buffer_store_b64(wX, wx, (int)sx, (int)(ts * ox_.rs * 2u), 2);
        __builtin_amdgcn_raw_buffer_store_b64(wZ, wz, (int)sz, (int)(ts * oz_.rs * 2u), 2);
        if (pad_lane) {
          const lean_u2 zz = {0u, 0u};
          __builtin_amdgcn_raw_buffer_store_b64(zz, wb, (int)spb, (int)(ts * ob_.rs * 2u), 2);
          __builtin_amdgcn_raw_buffer_store_b64(zz, wc, (int)spc, (int)(ts * oc_.rs * 2u), 2);
        }
      }
      // d delta of the head: the quad's 16 channels; token 4j+i's value parks in lane i of the quad, one store per four tokens
      float dq = quad_sum(qs) * LN2_F;
      if (d.softplus) dq *= 1.f - __builtin_amdgcn_exp2f(-dlv * LOG2E_F);      // sigmoid(x) = 1 - exp(-softplus(x))
      if (qi == i) ddl_keep = dq;
      __builtin_amdgcn_sched_barrier(0);   // (a token at a time, in this order: the whole walk is one basic block, and hipcc otherwise
                                           //  starts the gate arithmetic of later tokens early and holds its results: spills)
    }
    if (L.ok && 4 * j + qi < rows) ddp[(int64_t)(4 * j) * d.h] = ddl_keep;
  }
  // ---- the chunk's partial sums of dA_log and dD: the waves' in a fixed order ----
  sP[0][wv][ln] = make_float4(dA[0] * LN2_F, dA[1] * LN2_F, dA[2] * LN2_F, dA[3] * LN2_F);
  sP[1][wv][ln] = make_float4(dD[0], dD[1], dD[2], dD[3]);
  lb_barrier();
  if (wv == 0 && L.ok) {
    float4 a = sP[0][0][ln], dd = sP[1][0][ln];
#pragma unroll
    for (int w = 1; w < LB_NW; ++w) {
      const float4 a2 = sP[0][w][ln], d2 = sP[1][w][ln];
      a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
      dd.x += d2.x; dd.y += d2.y; dd.z += d2.z; dd.w += d2.w;
    }
    float *po = part + (((int64_t)L.b * d.nchunks + chunk) * 2) * d.Dn + L.c0;
    *reinterpret_cast<float4 *>(po) = a;
    *reinterpret_cast<float4 *>(po + d.Dn) = dd;
  }
}

// ---- host side ----
struct LbShape { ScanDims d; LbGeo G; int64_t T; };

int lb_shape(LbShape &s, int64_t B, int64_t L, int64_t h, int64_t N, int softplus,
             std::initializer_list<std::pair<const void *, int64_t>> slices, int64_t width_max) {
  int rc = make_dims(s.d, B, L, h, N, softplus);
  if (rc) return rc;
  s.d.nchunks = (int)ceil_div64(L, LB_LT);
  // N = 16: the four lanes of a head are a DPP quad.  Dn <= 256: a row fits a wave.
  if (s.d.N != 16 || s.d.Dn > 256) return APERTIS_ERR_UNSUPPORTED;
  s.T = B * L;
  LbGeo &G = s.G;
  G.g = (int)(s.d.Dn / 4);
  G.R = 64 / G.g;
  G.nbw = (int)ceil_div64(B, G.R);
  G.nsup = (int)ceil_div64(s.d.nchunks, LB_SUP);
  G.nck16 = (int)ceil_div64(L, LB_TW);
  int64_t rs_max = 0;
  for (auto &sl : slices) {
    if (sl.second < s.d.Dn) return APERTIS_ERR_ARG;
    if ((((uintptr_t)sl.first) | (uint64_t)(sl.second * 2)) & 7) return APERTIS_ERR_UNSUPPORTED;
    rs_max = std::max(rs_max, sl.second);
  }
  if ((s.T * rs_max + width_max) * 2 >= 0xfff00000LL || s.T * h * 4 >= 0xfff00000LL) return APERTIS_ERR_UNSUPPORTED;
  const int64_t bytesA = B * s.d.nchunks * 3 * G.g * 16, bytesI = B * G.nsup * 2 * G.g * 16;
  if (64 + bytesA + bytesI >= 0xfff00000LL || (int64_t)s.d.nchunks * G.nbw >= 0x7fffffffLL) return APERTIS_ERR_UNSUPPORTED;
  G.offA = 64u;
  G.offI = (uint32_t)(64 + bytesA);
  G.wsbytes = (uint32_t)(64 + bytesA + bytesI);
  return APERTIS_OK;
}

// one work-group per ticket (chunk-major)
int64_t lb_grid(const LbShape &s) {
  return (int64_t)s.d.nchunks * s.G.nbw;
}

}  // namespace


extern "C" int64_t apertis_scan_lookback_workspace_bytes(int64_t B, int64_t L, int64_t Dn) {
  if (B <= 0 || L <= 0 || Dn <= 0 || Dn % 4) return 0;
  const int64_t g = Dn / 4, nch = ceil_div64(L, LB_LT), nsup = ceil_div64(nch, LB_SUP);
  return 64 + B * nch * 3 * g * 16 + B * nsup * 2 * g * 16;
}

extern "C" int apertis_scan_lookback_fwd(const float *dlt, const float *A_log, const void *Bt, int64_t bt_rs, const void *C,
                                         int64_t c_rs, const void *xc, int64_t xc_rs, const void *z, int64_t z_rs, const float *D,
                                         const float *h0, void *out, int64_t out_rs, float *h_last, float *h_in, float *ckpt16,
                                         void *ws, uint32_t epoch, int64_t B, int64_t L, int64_t h, int64_t N, int delta_softplus,
                                         void *stream) {
  if (!dlt || !A_log || !Bt || !C || !xc || !z || !D || !out || !ws || epoch == 0) return APERTIS_ERR_ARG;
  LbShape s;
  int rc = lb_shape(s, B, L, h, N, delta_softplus, {{Bt, bt_rs}, {C, c_rs}, {xc, xc_rs}, {z, z_rs}, {out, out_rs}}, h * N);
  if (rc) return rc;
  if ((uintptr_t)ws & 15) return APERTIS_ERR_ARG;
  const int64_t Dn = s.d.Dn, T = s.T;
  auto lt = [&](const void *p, int64_t rs) { return LeanT{p, (uint32_t)rs, (uint32_t)(((T - 1) * rs + Dn) * 2)}; };
  const LeanT tdl{dlt, (uint32_t)h, (uint32_t)(T * h * 4)};
  const unsigned grid = (unsigned)lb_grid(s);
  hipLaunchKernelGGL(scan_lb_fwd_k, dim3(grid), dim3(64 * LB_NW), 0, (hipStream_t)stream, tdl, A_log, lt(Bt, bt_rs), lt(C, c_rs),
                     lt(xc, xc_rs), lt(z, z_rs), D, h0, h_in, h_last, ckpt16, lt(out, out_rs), (GateWsHead *)ws, epoch, s.d, s.G);
  return apertis_check_launch();
}

extern "C" int apertis_scan_lookback_bwd(const float *dlt, const float *A_log, const void *Bt, int64_t bt_rs, const void *C,
                                         int64_t c_rs, const void *xc, int64_t xc_rs, const void *z, int64_t z_rs, const float *D,
                                         const void *dout, int64_t dout_rs, const float *ckpt16, void *dBt, int64_t dbt_rs, void *dC,
                                         int64_t dc_rs, int64_t store_w, void *dxc, int64_t dxc_rs, void *dz, int64_t dz_rs,
                                         float *d_dlt, float *dA_dD, float *fold, float *part, void *ws, uint32_t epoch, int64_t B,
                                         int64_t L, int64_t h, int64_t N, int delta_softplus, void *stream) {
  if (!dlt || !A_log || !Bt || !C || !xc || !z || !D || !dout || !ckpt16 || !dBt || !dC || !dxc || !dz || !d_dlt || !dA_dD || !fold ||
      !part || !ws || epoch == 0)
    return APERTIS_ERR_ARG;
  LbShape s;
  int rc = lb_shape(s, B, L, h, N, delta_softplus,
                    {{Bt, bt_rs}, {C, c_rs}, {xc, xc_rs}, {z, z_rs}, {dout, dout_rs}, {dxc, dxc_rs}, {dz, dz_rs}, {dBt, dbt_rs}, {dC, dc_rs}},
                    store_w);
  if (rc) return rc;
  if ((uintptr_t)ws & 15) return APERTIS_ERR_ARG;
  const int64_t Dn = s.d.Dn, T = s.T;
  if (store_w < Dn || store_w > ceil_div64(Dn, TC) * TC || store_w % 4 || dbt_rs < store_w || dc_rs < store_w) return APERTIS_ERR_ARG;
  if ((store_w - Dn) / 4 > s.G.g) return APERTIS_ERR_UNSUPPORTED;     // (the pad columns are written by the row's own lanes)
  hipStream_t st = (hipStream_t)stream;
  auto lt = [&](const void *p, int64_t rs, int64_t w) { return LeanT{p, (uint32_t)rs, (uint32_t)(((T - 1) * rs + w) * 2)}; };
  const LeanT tdl{dlt, (uint32_t)h, (uint32_t)(T * h * 4)};
  const unsigned grid = (unsigned)lb_grid(s);
  hipLaunchKernelGGL(scan_lb_bwd_k, dim3(grid), dim3(64 * LB_NW), 0, st, tdl, A_log, lt(Bt, bt_rs, Dn), lt(C, c_rs, Dn), lt(xc, xc_rs, Dn),
                     lt(z, z_rs, Dn), lt(dout, dout_rs, Dn), D, ckpt16, lt(dBt, dbt_rs, store_w), lt(dC, dc_rs, store_w), (int)store_w,
                     lt(dxc, dxc_rs, Dn), lt(dz, dz_rs, Dn), d_dlt, part, (GateWsHead *)ws, epoch, s.d, s.G);
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
