// MoE routing for gfx950: gate softmax/top-k, the dispatch plan (histogram + capacity +
// radix-select + stable ranks), gather + per-expert LayerNorm, and the weighted combine.
//
// Reference: AdaptiveExpertSystem.forward, /root/reference/src/model/core.py:470-607.
// The reference walks a K x E Python loop with host syncs (nonzero / .any() / topk); here the
// whole plan is built on the device with no host round-trip.  Canonical row order is
// expert-major, then k, then ascending token - equivalent to the reference's k-major loop
// because capacity is consumed per expert.
//
// All of these are HBM/latency-bound integer and row-streaming kernels: one wave per row with
// 8/16-byte vector accesses, integer atomics only (deterministic), float atomics only for the
// LayerNorm affine gradients.
#include "common.h"

namespace {

constexpr int MAXE = 64;  // experts
constexpr int MAXK = 8;   // experts per token

// ------------------------------------------------------------------------------------------
// gate: softmax -> top-K (ties: lowest expert index) -> renormalised weights
// ------------------------------------------------------------------------------------------
template <int EC>  // EC > 0: compile-time expert count; EC == 0: runtime E <= MAXE
__device__ __forceinline__ void gate_topk_row(const float *__restrict__ row, float *__restrict__ gates_row, int32_t *idx_row,
                                              float *w_row, int E_rt, int K) {
  constexpr int CAP = EC > 0 ? EC : MAXE;
  const int E = EC > 0 ? EC : E_rt;
  float v[CAP];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < CAP; ++i)
    if (i < E) { v[i] = row[i]; m = fmaxf(m, v[i]); }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < CAP; ++i)
    if (i < E) { v[i] = expf(v[i] - m); sum += v[i]; }
#pragma unroll
  for (int i = 0; i < CAP; ++i)
    if (i < E) { v[i] = v[i] / sum; gates_row[i] = v[i]; }
  uint64_t chosen = 0;
  float p[MAXK];
  float psum = 0.f;
#pragma unroll
  for (int k = 0; k < MAXK; ++k) {
    if (k < K) {
      float best = -1.f;
      int bi = 0;
#pragma unroll
      for (int i = 0; i < CAP; ++i)
        if (i < E && !((chosen >> i) & 1) && v[i] > best) { best = v[i]; bi = i; }
      chosen |= 1ull << bi;
      idx_row[k] = bi;
      p[k] = best;
      psum += best;
    }
  }
  const float den = psum + 1e-6f;  // core.py:529
#pragma unroll
  for (int k = 0; k < MAXK; ++k)
    if (k < K) w_row[k] = p[k] / den;
}

template <int EC>
__global__ void gate_topk_fwd_k(const float *__restrict__ logits, float *__restrict__ gates,
                                int32_t *__restrict__ idx, float *__restrict__ w, int64_t S, int E_rt,
                                int K) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  const int E = EC > 0 ? EC : E_rt;
  gate_topk_row<EC>(logits + s * E, gates + s * E, idx + s * K, w + s * K, E_rt, K);
}

// dlogits from dw (through renorm + top-k gather) and dgates (aux losses), softmax backward
template <int EC>
__global__ void gate_topk_bwd_k(const float *__restrict__ gates, const int32_t *__restrict__ idx,
                                const float *__restrict__ dw, const float *__restrict__ dgates,
                                float *__restrict__ dlogits, int64_t S, int E_rt, int K) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  constexpr int CAP = EC > 0 ? EC : MAXE;
  const int E = EC > 0 ? EC : E_rt;
  float g[CAP], dg[CAP];
#pragma unroll
  for (int i = 0; i < CAP; ++i)
    if (i < E) { g[i] = gates[s * E + i]; dg[i] = dgates ? dgates[s * E + i] : 0.f; }
  if (dw) {
    float psum = 0.f, dot = 0.f;
    for (int k = 0; k < K; ++k) {
      int e = idx[s * K + k];
      float pe = 0.f;
#pragma unroll
      for (int i = 0; i < CAP; ++i)
        if (i == e) pe = g[i];
      psum += pe;
      dot += dw[s * K + k] * pe;
    }
    const float den = psum + 1e-6f;
    const float corr = dot / (den * den);
    for (int k = 0; k < K; ++k) {
      int e = idx[s * K + k];
      float dp = dw[s * K + k] / den - corr;
#pragma unroll
      for (int i = 0; i < CAP; ++i)
        if (i == e) dg[i] += dp;
    }
  }
  float inner = 0.f;
#pragma unroll
  for (int i = 0; i < CAP; ++i)
    if (i < E) inner += dg[i] * g[i];
#pragma unroll
  for (int i = 0; i < CAP; ++i)
    if (i < E) dlogits[s * E + i] = g[i] * (dg[i] - inner);
}

// ------------------------------------------------------------------------------------------
// dispatch plan
// ------------------------------------------------------------------------------------------
constexpr int PLAN_HIST_BLOCKS = 512;   // most work-groups of the candidate histogram (each leaves one row of counts)
struct PlanWs {
  int32_t *total, *keep, *mode, *quota, *seg_start, *bpart;   // bpart [PLAN_HIST_BLOCKS][P]: the histogram launch's rows
  uint32_t *thr, *smask;   // radix select: prefix found so far / bits already fixed
  int32_t *ghist;          // [P][256] digit histogram of the current radix pass
  int32_t *done;           // [4] work-groups of plan_sel_hist_k that have added their bins, per digit (right behind ghist: one memset)
  int32_t *cnt_g, *cnt_t;  // [P][NCH]
  int P, NCH;
};

PlanWs carve_ws(void *ws, int64_t S, int64_t E, int64_t K) {
  PlanWs w;
  w.P = (int)(E * K);
  w.NCH = (int)ceil_div64(S, 64);
  int32_t *p = (int32_t *)ws;
  w.total = p; p += w.P;
  w.keep = p; p += w.P;
  w.mode = p; p += w.P;
  w.quota = p; p += w.P;
  w.seg_start = p; p += w.P;
  w.thr = (uint32_t *)p; p += w.P;
  w.smask = (uint32_t *)p; p += w.P;
  w.ghist = p; p += (int64_t)w.P * 256;
  w.done = p; p += 4;
  w.cnt_g = p; p += (int64_t)w.P * w.NCH;
  w.cnt_t = p; p += (int64_t)w.P * w.NCH;
  w.bpart = p;   // [PLAN_HIST_BLOCKS][P]
  return w;
}

// Candidates per (expert, k) slot.  Every work-group leaves ONE row of P counts in `bpart` (no global atomics: with the first
// form's atomicAdd per block and slot, 512-1024 work-groups queued up on the same 16 addresses - 41 us for 1 MB of indices on
// the H = 256 configuration); inside a work-group a wave counts each slot it holds with one ballot (a 64-lane LDS atomic on
// <= 16 addresses serialises).  plan_capacity_k sums the rows in block order.
__global__ void __launch_bounds__(256)
plan_hist_k(const int32_t *__restrict__ idx, int32_t *__restrict__ bpart, int64_t SK, int E, int K) {
  __shared__ int32_t h[MAXE * MAXK];
  const int P = E * K, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < P; i += blockDim.x) h[i] = 0;
  __syncthreads();
  for (int64_t a0 = (int64_t)blockIdx.x * blockDim.x; a0 < SK; a0 += (int64_t)gridDim.x * blockDim.x) {   // (uniform trip count)
    const int64_t a = a0 + threadIdx.x;
    int slot = -1;
    if (a < SK) {
      const int e = idx[a];
      if (e >= 0 && e < E) slot = e * K + (int)(a % K);
    }
    unsigned long long todo = __ballot(slot >= 0);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int s0 = __shfl(slot, leader);
      const unsigned long long m = __ballot(slot == s0);
      if (lane == leader) atomicAdd(&h[s0], __popcll(m));
      todo &= ~m;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < P; i += blockDim.x) bpart[(int64_t)blockIdx.x * P + i] = h[i];
}

// per expert: consume capacity k-major (core.py:547-576); then lay the segments out expert-major
__global__ void plan_capacity_k(PlanWs w, const uint8_t *__restrict__ active, int64_t capacity,
                                int32_t *__restrict__ offsets, int E, int K, int nhist) {
  // totals: the histogram launch's per-work-group rows (integers: any order gives the same sums).  256 threads, 256 / P of them
  // per slot when the slots are few
  __shared__ int32_t s_sum[256];
  {
    const int P = E * K;
    const int g = P < 256 ? 256 / P : 1;                 // threads per slot
    for (int p0 = 0; p0 < P; p0 += 256 / g) {
      const int p = p0 + (int)threadIdx.x / g, sub = (int)threadIdx.x % g;
      int t = 0;
      if (p < P && (int)threadIdx.x / g < 256 / g)
        for (int b = sub; b < nhist; b += g) t += w.bpart[(int64_t)b * P + p];
      s_sum[threadIdx.x] = t;
      __syncthreads();
      if (p < P && sub == 0 && (int)threadIdx.x / g < 256 / g) {
        int tt = 0;
        for (int q = 0; q < g; ++q) tt += s_sum[threadIdx.x + q];
        w.total[p] = tt;
      }
      __syncthreads();
    }
  }
  const int e = threadIdx.x;
  if (e < E) {
    int64_t load = 0;
    const bool on = active ? active[e] != 0 : true;
    for (int k = 0; k < K; ++k) {
      const int p = e * K + k;
      const int tot = w.total[p];
      int64_t keep = tot;
      if (!on) keep = 0;
      else if (capacity > 0) {
        int64_t rem = capacity - load;
        keep = rem <= 0 ? 0 : (tot < rem ? tot : rem);
      }
      w.keep[p] = (int)keep;
      w.mode[p] = keep == 0 ? 0 : (keep == tot ? 1 : 2);
      w.quota[p] = (int)keep;   // radix select: how many of the slot's candidates are still to be taken
      w.thr[p] = 0;
      w.smask[p] = 0;
      load += keep;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int ee = 0; ee < E; ++ee) {
      offsets[ee] = run;
      for (int k = 0; k < K; ++k) {
        w.seg_start[ee * K + k] = run;
        run += w.keep[ee * K + k];
      }
    }
    offsets[E] = run;
  }
}

// overflowing (e,k) slot: find the keep-th largest gate weight T (radix select on the float
// bits; weights are >= 0 so unsigned order == float order) and how many ties at T to keep
__global__ void __launch_bounds__(1024)
plan_select_k(PlanWs w, const int32_t *__restrict__ idx, const float *__restrict__ wk, int64_t S, int K) {
  const int p = blockIdx.x;
  if (w.mode[p] != 2) return;
  const int e = p / K, k = p - e * K;
  __shared__ int32_t hist[256];
  __shared__ uint32_t s_prefix, s_mask;
  __shared__ int32_t s_need;
  if (threadIdx.x == 0) { s_prefix = 0; s_mask = 0; s_need = w.keep[p]; }
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const uint32_t prefix = s_prefix, mask = s_mask;
    // 8 tokens per thread and trip, all 16 loads issued before the first use: one block per (e,k) walks
    // all S tokens four times, and with one dependent load pair per trip that walk was pure latency (200 us)
    for (int64_t s0 = threadIdx.x; s0 < S; s0 += (int64_t)blockDim.x * 8) {
      int32_t ei[8];
      uint32_t bi[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t s = s0 + (int64_t)u * blockDim.x;
        ei[u] = s < S ? idx[s * K + k] : -1;
        bi[u] = s < S ? __float_as_uint(wk[s * K + k]) : 0u;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (ei[u] == e && (bi[u] & mask) == prefix) atomicAdd(&hist[(bi[u] >> shift) & 255], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int need = s_need, cum = 0;
      for (int b = 255; b >= 0; --b) {
        if (cum + hist[b] >= need) {
          s_need = need - cum;
          s_prefix = prefix | ((uint32_t)b << shift);
          s_mask = mask | (255u << shift);
          break;
        }
        cum += hist[b];
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { w.thr[p] = s_prefix; w.quota[p] = s_need; }
}

// one wave per slot: the bucket (from the top) where the running count reaches the slot's remaining need.  Runs in the LAST
// work-group of plan_sel_hist_k to finish (round 6: a launch less per digit, four per plan): the bins were added by other
// work-groups with device-scope atomics and are read here with device-scope loads (the XCDs' L2s are not coherent inside a kernel).
__device__ __forceinline__ void plan_sel_pick(const PlanWs &w, int p, int lane, int shift) {
  if (w.mode[p] != 2) return;
  int32_t *h = w.ghist + p * 256;
  // lane l owns bins 255-4l .. 252-4l (descending)
  int c[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) c[j] = __hip_atomic_load(h + 255 - 4 * lane - j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int mine = (c[0] + c[1]) + (c[2] + c[3]);
  int incl = mine;   // inclusive prefix over lanes 0..l
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  const int need = w.quota[p];
  const int before = incl - mine;
  if (before < need && incl >= need) {   // exactly one lane
    int cum = before, b = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (cum < need && cum + c[j] >= need) { b = 255 - 4 * lane - j; break; }
      cum += c[j];
    }
    w.quota[p] = need - cum;
    w.thr[p] = w.thr[p] | ((uint32_t)b << shift);
    w.smask[p] = w.smask[p] | (255u << shift);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) h[255 - 4 * lane - j] = 0;   // ready for the next digit
}

// The same radix select spread over the chip (one launch pair per 8-bit digit): every work-group histograms
// its share of the tokens for ALL overflowing slots in LDS and adds the non-empty bins to the global
// histogram (integer atomics: exact, order-free); the last work-group to finish then fixes the digit per slot (plan_sel_pick).  The
// one-block-per-slot kernel above walks all S tokens four times with at most E*K blocks busy (154 us at
// S = 131k with 8 overflowing slots); kept for P > 32 slots (LDS).
__global__ void __launch_bounds__(256)
plan_sel_hist_k(PlanWs w, const int32_t *__restrict__ idx, const float *__restrict__ wk, int64_t S, int E, int K, int shift, int digit) {
  extern __shared__ int32_t lh[];   // [P][256]
  __shared__ int32_t s_mode[32];     // (P <= 32 on this path) the slots' state, once per work-group instead of three dependent
  __shared__ uint32_t s_mask[32], s_thr[32];   // global loads per element
  const int P = w.P;
  for (int i = threadIdx.x; i < P * 256; i += 256) lh[i] = 0;
  if (threadIdx.x < P) { s_mode[threadIdx.x] = w.mode[threadIdx.x]; s_mask[threadIdx.x] = w.smask[threadIdx.x]; s_thr[threadIdx.x] = w.thr[threadIdx.x]; }
  __syncthreads();
  const int64_t SK = S * K;
  for (int64_t a0 = (int64_t)blockIdx.x * 256 + threadIdx.x; a0 < SK; a0 += (int64_t)gridDim.x * 256 * 4) {
    int32_t ei[4];
    uint32_t bi[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t a = a0 + (int64_t)u * gridDim.x * 256;
      ei[u] = a < SK ? idx[a] : -1;
      bi[u] = a < SK ? __float_as_uint(wk[a]) : 0u;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t a = a0 + (int64_t)u * gridDim.x * 256;
      if (ei[u] >= 0 && ei[u] < E) {
        const int p = ei[u] * K + (int)(a % K);
        if (s_mode[p] == 2 && (bi[u] & s_mask[p]) == s_thr[p]) atomicAdd(&lh[p * 256 + ((bi[u] >> shift) & 255)], 1);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < P * 256; i += 256)
    if (lh[i]) atomicAdd(&w.ghist[i], lh[i]);
  // the last work-group to get here fixes the digit of every slot (round 5: a launch of its own per digit)
  __threadfence();
  __syncthreads();
  __shared__ int s_last;
  if (threadIdx.x == 0) s_last = atomicAdd(&w.done[digit], 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  for (int p = threadIdx.x >> 6; p < P; p += 4) plan_sel_pick(w, p, threadIdx.x & 63, shift);
}

__device__ __forceinline__ void plan_flags(const PlanWs &w, const int32_t *idx, const float *wk, int64_t s,
                                           int64_t S, int k, int E, int K, int &e, bool &fg, bool &ft) {
  e = -1; fg = false; ft = false;
  if (s < S) {
    e = idx[s * K + k];
    if (e >= 0 && e < E) {
      const int p = e * K + k;
      const int m = w.mode[p];
      if (m == 1) fg = true;
      else if (m == 2) {
        uint32_t bits = __float_as_uint(wk[s * K + k]);
        uint32_t t = w.thr[p];
        fg = bits > t;
        ft = bits == t;
      }
    } else e = -1;
  }
}

// per 64-token chunk and (e,k): number of kept-for-sure rows and of threshold ties
__global__ void plan_count_k(PlanWs w, const int32_t *__restrict__ idx, const float *__restrict__ wk,
                             int64_t S, int E, int K) {
  const int lane = threadIdx.x & 63;
  const int64_t chunk = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (chunk >= w.NCH) return;
  const int64_t s = chunk * 64 + lane;
  for (int k = 0; k < K; ++k) {
    int e; bool fg, ft;
    plan_flags(w, idx, wk, s, S, k, E, K, e, fg, ft);
    for (int ee = 0; ee < E; ++ee) {
      unsigned long long bg = __ballot(e == ee && fg);
      unsigned long long bt = __ballot(e == ee && ft);
      if (lane == 0) {
        w.cnt_g[(int64_t)(ee * K + k) * w.NCH + chunk] = __popcll(bg);
        w.cnt_t[(int64_t)(ee * K + k) * w.NCH + chunk] = __popcll(bt);
      }
    }
  }
}

// in-place exclusive scan of 2P arrays of NCH ints (blockIdx.x picks the array)
__global__ void __launch_bounds__(256) plan_scan_k(PlanWs w) {
  int32_t *arr = (blockIdx.x < (unsigned)w.P ? w.cnt_g + (int64_t)blockIdx.x * w.NCH
                                              : w.cnt_t + (int64_t)(blockIdx.x - w.P) * w.NCH);
  __shared__ int32_t part[256];
  const int per = (w.NCH + 255) / 256;
  const int b0 = threadIdx.x * per, b1 = min(b0 + per, w.NCH);
  int sum = 0;
  for (int i = b0; i < b1; ++i) sum += arr[i];
  part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i < 256; ++i) { int t = part[i]; part[i] = run; run += t; }
  }
  __syncthreads();
  int run = part[threadIdx.x];
  for (int i = b0; i < b1; ++i) { int t = arr[i]; arr[i] = run; run += t; }
}

__global__ void plan_assign_k(PlanWs w, const int32_t *__restrict__ idx, const float *__restrict__ wk,
                              int32_t *__restrict__ row_token, int32_t *__restrict__ row_k,
                              int32_t *__restrict__ slot_of, int64_t S, int E, int K) {
  const int lane = threadIdx.x & 63;
  const int64_t chunk = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (chunk >= w.NCH) return;
  const int64_t s = chunk * 64 + lane;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int k = 0; k < K; ++k) {
    int e; bool fg, ft;
    plan_flags(w, idx, wk, s, S, k, E, K, e, fg, ft);
    int pre_g = 0, pre_t = 0;
    for (int ee = 0; ee < E; ++ee) {
      unsigned long long bg = __ballot(e == ee && fg);
      unsigned long long bt = __ballot(e == ee && ft);
      if (e == ee) { pre_g = __popcll(bg & lt); pre_t = __popcll(bt & lt); }
    }
    if (s < S) {
      int slot = -1;
      if (e >= 0 && (fg || ft)) {
        const int p = e * K + k;
        const int gb = w.cnt_g[(int64_t)p * w.NCH + chunk] + pre_g;
        const int tb = w.cnt_t[(int64_t)p * w.NCH + chunk] + pre_t;
        const int q = w.quota[p];
        if (fg || tb < q) {
          slot = w.seg_start[p] + gb + (tb < q ? tb : q);
          row_token[slot] = (int32_t)s;
          row_k[slot] = k;
        }
      }
      slot_of[s * K + k] = slot;
    }
  }
}

// The whole plan in ONE launch for a handful of tokens (S <= 64, E * K <= 16: the single-token decode step, reference
// core.py:1578-1603 - the five launches above were a fifth of a captured token step).  One wave per (expert, k) slot,
// lane = token.  Same semantics: candidates idx[s, k] == e; per expert the capacity is consumed k-major; an overflowing
// slot keeps its `keep` largest gate weights (compared as bits, as the radix select does), ties at the threshold in token
// order; the kept rows of a slot sit in token order, the slots expert-major then k.
__device__ __forceinline__ void
plan_small_body(const int32_t *idx, const float *wk, const uint8_t *__restrict__ active, int64_t capacity,
                int32_t *__restrict__ offsets, int32_t *__restrict__ row_token, int32_t *__restrict__ row_k,
                int32_t *__restrict__ slot_of, int S, int E, int K, int32_t *s_off, int32_t *s_rtok) {
  // (s_off [E + 1] / s_rtok [S * K]: optional LDS copies of offsets / row_token for a caller that carries on in the same launch)
  __shared__ int32_t s_tot[16], s_keep[16], s_start[16];
  const int P = E * K, p = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
  const int e = p / K, k = p - e * K;
  int es = -1;
  uint32_t bits = 0;
  if (p < P && lane < S) {
    es = idx[lane * K + k];
    bits = __float_as_uint(wk[lane * K + k]);
  }
  const bool cand = p < P && es == e;
  const unsigned long long cm = __ballot(cand);
  if (p < P && lane == 0) s_tot[p] = __popcll(cm);
  __syncthreads();
  if ((int)threadIdx.x < E) {
    const int ee = threadIdx.x;
    const bool on = active ? active[ee] != 0 : true;
    int64_t load = 0;
    for (int kk = 0; kk < K; ++kk) {
      const int tot = s_tot[ee * K + kk];
      int64_t keep = tot;
      if (!on) keep = 0;
      else if (capacity > 0) {
        const int64_t rem = capacity - load;
        keep = rem <= 0 ? 0 : (tot < rem ? tot : rem);
      }
      s_keep[ee * K + kk] = (int)keep;
      load += keep;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int ee = 0; ee < E; ++ee) {
      offsets[ee] = run;
      if (s_off) s_off[ee] = run;
      for (int kk = 0; kk < K; ++kk) { s_start[ee * K + kk] = run; run += s_keep[ee * K + kk]; }
    }
    offsets[E] = run;
    if (s_off) s_off[E] = run;
  }
  __syncthreads();
  if (p >= P) return;
  const int keep = s_keep[p], tot = s_tot[p];
  bool kept = cand && keep > 0;
  if (keep > 0 && keep < tot) {      // overflow: rank among the slot's candidates by (weight descending, token ascending)
    int pos = 0;
    for (int j = 0; j < S; ++j) {
      const uint32_t bj = (uint32_t)__shfl((int)bits, j);
      if ((cm >> j) & 1ull) pos += (bj > bits) || (bj == bits && j < lane);
    }
    kept = cand && pos < keep;
  }
  const unsigned long long km = __ballot(kept);
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  if (lane < S) {
    if (cand) {
      int slot = -1;
      if (kept) {
        slot = s_start[p] + __popcll(km & lt);
        row_token[slot] = lane;
        row_k[slot] = k;
        if (s_rtok) s_rtok[slot] = lane;
      }
      slot_of[lane * K + k] = slot;
    } else if (e == 0 && (es < 0 || es >= E)) {
      slot_of[lane * K + k] = -1;     // an index outside [0, E): no expert's wave claims the pair
    }
  }
}

__global__ void __launch_bounds__(1024)
plan_small_k(const int32_t *__restrict__ idx, const float *__restrict__ wk, const uint8_t *__restrict__ active, int64_t capacity,
             int32_t *__restrict__ offsets, int32_t *__restrict__ row_token, int32_t *__restrict__ row_k,
             int32_t *__restrict__ slot_of, int S, int E, int K) {
  plan_small_body(idx, wk, active, capacity, offsets, row_token, row_k, slot_of, S, E, K, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------------
// row helpers: a wave owns one row of H elements, lane handles 4-element chunks lane+64*i
// ------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float4 load4(const T *p);
template <> __device__ __forceinline__ float4 load4<float>(const float *p) {
  return *reinterpret_cast<const float4 *>(p);
}
template <> __device__ __forceinline__ float4 load4<bf16_t>(const bf16_t *p) {
  uint2 u = *reinterpret_cast<const uint2 *>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                     __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
// Streaming row data, read once and written once: NON-TEMPORAL both ways.  Every tensor these kernels read or write is 0.2 - 0.5
// GB, larger than what the caches can hand from producer to consumer; moved through them it displaces what the next kernels
// read.  Measured on the whole step (A/B inside one gpurun call, sums of kernel times): `nt` stores in the row kernels alone
// -7 ms per step, most of it in the expert GEMMs that FOLLOW them (they run 1.5 - 2.5 % faster); with the GEMM epilogues',
// the scan outputs' and AdamW's accesses non-temporal as well 479.6 -> 468.9 ms.  Which kernel gains depends on its
// neighbours (the LayerNorm backward is 2 % slower with nt stores, the combine backward behind it 15 % faster), so the
// choice was made on the step, not per kernel.  The affine vectors (gamma, beta, W) stay on plain loads: they are re-read.
template <typename T> __device__ __forceinline__ float4 load4s(const T *p);
template <> __device__ __forceinline__ float4 load4s<float>(const float *p) {
  typedef __attribute__((ext_vector_type(4))) float f4;
  const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p));
  return make_float4(t.x, t.y, t.z, t.w);
}
template <> __device__ __forceinline__ float4 load4s<bf16_t>(const bf16_t *p) {
  typedef __attribute__((ext_vector_type(2))) unsigned u2;
  const u2 u = __builtin_nontemporal_load(reinterpret_cast<const u2 *>(p));
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                     __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
template <typename T> __device__ __forceinline__ void store4(T *p, float4 v);
// (what the write-heavy kernels spend on their stores - round 5, a probe build with the data stores compiled out:
//  profiles/r5_probe_row_kernels_nostore.log; the switch left this file in round 6, round 5's tree has it - tools/probes/README.md)
template <> __device__ __forceinline__ void store4<float>(float *p, float4 v) {
  typedef __attribute__((ext_vector_type(4))) float f4;
  f4 o = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(o, reinterpret_cast<f4 *>(p));
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t *p, float4 v) {
  typedef __attribute__((ext_vector_type(4))) bf16_t bf4;
  typedef __attribute__((ext_vector_type(2))) unsigned u2;
  bf4 o = {(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
  __builtin_nontemporal_store(__builtin_bit_cast(u2, o), reinterpret_cast<u2 *>(p));
}

// a row chunk kept in its storage form (the persistent row kernels hold the NEXT row this way: half the registers for bf16)
template <typename TX> struct raw4;
template <> struct raw4<float> { typedef float4 type; };
template <> struct raw4<bf16_t> { typedef uint2 type; };
// a row chunk in its storage form, streamed (non-temporal, see load4s)
__device__ __forceinline__ float4 raw_load(const float *p) {
  typedef __attribute__((ext_vector_type(4))) float f4;
  const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p));
  return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ uint2 raw_load(const bf16_t *p) {
  typedef __attribute__((ext_vector_type(2))) unsigned u2;
  const u2 t = __builtin_nontemporal_load(reinterpret_cast<const u2 *>(p));
  return make_uint2(t.x, t.y);
}
__device__ __forceinline__ float4 raw_to_f4(const float4 &v) { return v; }
__device__ __forceinline__ float4 raw_to_f4(const uint2 &u) {
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}

// Sum over the 64 lanes, the same value in every lane.  DPP adds inside the rows of 16 (quad swaps, half-row and row
// mirrors), row broadcasts across them, one v_readlane of lane 63: seven VALU instructions.  As six __shfl_xor steps
// (ds_bpermute_b32 + s_waitcnt lgkmcnt + add each, ~60 cycles of dependent latency per step) the ten reductions per row of
// the router forward were most of that kernel.  Fixed order: deterministic.
// 1 / H once per kernel (the compiler hoists it): the statistics of a row are sums TIMES this instead of sums divided by H -
// an IEEE division is ~10 instructions, and the row kernels did two to four of them per row
__device__ __forceinline__ float inv_h(int H) { return 1.f / (float)H; }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
  v += dpp_f<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
  v += dpp_f<0x141, 0xf>(v);   // row_half_mirror
  v += dpp_f<0x140, 0xf>(v);   // row_mirror: every lane holds its row's sum
  v += dpp_f<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  v += dpp_f<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3: lane 63 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ int expert_of_row(const int32_t *offsets, int E, int r) {
  int e = 0;
  while (e + 1 < E && offsets[e + 1] <= r) ++e;
  return e;
}

// xg[r,:] = LayerNorm(x[row_token[r],:]) * gamma[e] + beta[e]   (IT chunks of 4 per lane)
template <typename TX, typename TO, int IT>
__global__ void __launch_bounds__(256)
gather_ln_fwd_k(const TX *__restrict__ x, const int32_t *__restrict__ row_token,
                const int32_t *__restrict__ offsets, const float *__restrict__ gamma,
                const float *__restrict__ beta, float eps, TO *__restrict__ xg,
                float *__restrict__ mean_o, float *__restrict__ rstd_o, int64_t max_rows, int H, int E) {
  // the dispatch picks IT = ceil(H / 256) for IT <= 4: every chunk below the last lies inside the row, no bounds test needed
  if constexpr (IT <= 4) __builtin_assume(H > 256 * (IT - 1));
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= max_rows || (offsets && r >= offsets[E])) return;
  const int e = offsets ? expert_of_row(offsets, E, (int)r) : 0;
  const TX *src = x + (row_token ? (int64_t)row_token[r] : r) * H;   // row_token == NULL: plain LayerNorm
  float4 v[IT];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    int c = (lane + 64 * i) * 4;
    v[i] = c < H ? load4s<TX>(src + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(sum) * inv_h(H);
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    int c = (lane + 64 * i) * 4;
    if (c < H) {
      float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      sq += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) * inv_h(H) + eps);
  const float *ga = gamma + (int64_t)e * H, *be = beta + (int64_t)e * H;
  TO *dst = xg + r * H;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    int c = (lane + 64 * i) * 4;
    if (c < H) {
      float4 g4 = load4<float>(ga + c), b4 = load4<float>(be + c);
      float4 o = make_float4((v[i].x - mean) * rstd * g4.x + b4.x, (v[i].y - mean) * rstd * g4.y + b4.y,
                             (v[i].z - mean) * rstd * g4.z + b4.z, (v[i].w - mean) * rstd * g4.w + b4.w);
      store4<TO>(dst + c, o);
    }
  }
  if (lane == 0) { mean_o[r] = mean; rstd_o[r] = rstd; }
}

// ------------------------------------------------------------------------------------------
// A handful of tokens (the single-token decode step, reference core.py:1578-1603: S = batch <= 64 rows): gate, dispatch plan
// and gather-LayerNorm in ONE launch - as three they are three dependent 3-5 us kernels per layer of a token step that
// together move a few KB.  One work-group of E*K waves (plan_small_body's shape): threads < S run the gate (gate_topk_row:
// the same arithmetic as apertis_moe_gate_topk_fwd), idx / w go through LDS into the plan, whose offsets / row_token stay in
// LDS for the rows' LayerNorm (gather_ln_fwd_k's arithmetic, a wave per row).  Eval mode: no capacity, no dropped experts.
// ------------------------------------------------------------------------------------------
template <typename TX, typename TO, int IT, int EC>
__global__ void __launch_bounds__(1024)
moe_route_small_k(const float *__restrict__ logits, float *__restrict__ gates, int32_t *__restrict__ idx_o, float *__restrict__ w_o,
                  int32_t *__restrict__ offsets, int32_t *__restrict__ row_token, int32_t *__restrict__ row_k,
                  int32_t *__restrict__ slot_of, const TX *__restrict__ x, const float *__restrict__ gamma,
                  const float *__restrict__ beta, float eps, TO *__restrict__ xg, float *__restrict__ mean_o,
                  float *__restrict__ rstd_o, int S, int E, int K, int H) {
  if constexpr (IT <= 4) __builtin_assume(H > 256 * (IT - 1));
  __shared__ int32_t s_idx[64 * MAXK], s_off[17], s_rtok[64 * MAXK];
  __shared__ float s_w[64 * MAXK];
  const int t = (int)threadIdx.x;
  if (t < S) {
    gate_topk_row<EC>(logits + (int64_t)t * E, gates + (int64_t)t * E, s_idx + t * K, s_w + t * K, E, K);
    for (int k = 0; k < K; ++k) { idx_o[t * K + k] = s_idx[t * K + k]; w_o[t * K + k] = s_w[t * K + k]; }
  }
  __syncthreads();
  plan_small_body(s_idx, s_w, nullptr, 0, offsets, row_token, row_k, slot_of, S, E, K, s_off, s_rtok);
  __syncthreads();
  const int lane = t & 63, wave = t >> 6, nwaves = (int)blockDim.x >> 6, rows = s_off[E];
  for (int r = wave; r < rows; r += nwaves) {
    const int e = expert_of_row(s_off, E, r);
    const TX *src = x + (int64_t)s_rtok[r] * H;
    float4 v[IT];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int c = (lane + 64 * i) * 4;
      v[i] = c < H ? load4s<TX>(src + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(sum) * inv_h(H);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int c = (lane + 64 * i) * 4;
      if (c < H) {
        float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
        sq += (a * a + b * b) + (cc * cc + d * d);
      }
    }
    const float rstd = rsqrtf(wave_sum(sq) * inv_h(H) + eps);
    const float *ga = gamma + (int64_t)e * H, *be = beta + (int64_t)e * H;
    TO *dst = xg + (int64_t)r * H;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int c = (lane + 64 * i) * 4;
      if (c < H) {
        float4 g4 = load4<float>(ga + c), b4 = load4<float>(be + c);
        float4 o = make_float4((v[i].x - mean) * rstd * g4.x + b4.x, (v[i].y - mean) * rstd * g4.y + b4.y,
                               (v[i].z - mean) * rstd * g4.z + b4.z, (v[i].w - mean) * rstd * g4.w + b4.w);
        store4<TO>(dst + c, o);
      }
    }
    if (lane == 0) { mean_o[r] = mean; rstd_o[r] = rstd; }
  }
}

// Gather-LayerNorm backward per row, structured like layernorm_bwd_k (8 rows per wave, two rows in
// flight, 16 waves per CU; a one-row-at-a-time loop was latency-bound: 180 us for 41k rows of 704).  Affine gradients: per-wave register sums are
// flushed with float atomics when the expert changes inside the wave's rows (rare: rows are
// expert-sorted); at the end the block's four waves are combined in LDS first when they all ended
// in the same expert, so the common case issues one set of atomics per block of 4 * GLN_RPW rows.
constexpr int GLN_RPW = 16;   // rows per wave of the gather-LN backward (4 waves per block, one partial row per block)
template <typename TX, typename TG, int IT>
__global__ void __launch_bounds__(256, IT <= 3 ? 4 : 1)   // narrow rows: <= 128 VGPRs = four waves per SIMD (IT = 3 took 132)
gather_ln_bwd2_k(const TX *__restrict__ x, const int32_t *__restrict__ row_token, const int32_t *__restrict__ offsets,
                 const float *__restrict__ gamma, const float *__restrict__ mean_i, const float *__restrict__ rstd_i,
                 const TG *__restrict__ dxg, TG *__restrict__ dxr, float *__restrict__ dgamma, float *__restrict__ dbeta,
                 float *__restrict__ part, int32_t *__restrict__ blk_expert, int64_t max_rows, int H, int E) {
  // the dispatch picks IT = ceil(H / 256) for IT <= 4: every chunk below the last lies inside the row, no bounds test needed
  if constexpr (IT <= 4) __builtin_assume(H > 256 * (IT - 1));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4 *red = reinterpret_cast<float4 *>(smem);          // [3 waves][2][H/4]
  __shared__ int s_e[4];
  constexpr int RPW = GLN_RPW;
  bool flushed = false;   // this wave crossed an expert boundary and used atomics
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t total = min((int64_t)offsets[E], max_rows);
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + wv) * RPW, r1 = min(r0 + RPW, total);
  float4 ag[IT], ab[IT], g4[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) { ag[i] = make_float4(0, 0, 0, 0); ab[i] = make_float4(0, 0, 0, 0); g4[i] = make_float4(0, 0, 0, 0); }
  int e = -1;
  auto load_gamma = [&](int ee) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int c = (lane + 64 * i) * 4;
      if (c < H) g4[i] = load4<float>(gamma + (int64_t)ee * H + c);
    }
  };
  auto flush_atomic = [&](int ee) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int c = (lane + 64 * i) * 4;
      if (c < H) {
        float *dg = dgamma + (int64_t)ee * H + c, *db = dbeta + (int64_t)ee * H + c;
        atomicAdd(dg + 0, ag[i].x); atomicAdd(dg + 1, ag[i].y); atomicAdd(dg + 2, ag[i].z); atomicAdd(dg + 3, ag[i].w);
        atomicAdd(db + 0, ab[i].x); atomicAdd(db + 1, ab[i].y); atomicAdd(db + 2, ab[i].z); atomicAdd(db + 3, ab[i].w);
      }
      ag[i] = make_float4(0, 0, 0, 0); ab[i] = make_float4(0, 0, 0, 0);
    }
  };
  if (r0 < total) { e = expert_of_row(offsets, E, (int)r0); load_gamma(e); }
  for (int64_t r = r0; r < r1; r += 2) {
    const bool two = r + 1 < r1;
    float4 xv[2][IT], dv[2][IT];
    float mean[2], rstd[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int64_t rr = (q == 0 || two) ? r + q : r;
      mean[q] = mean_i[rr]; rstd[q] = rstd_i[rr];
      const TX *src = x + (int64_t)row_token[rr] * H;
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        int c = (lane + 64 * i) * 4;
        if (c < H) { xv[q][i] = load4s<TX>(src + c); dv[q][i] = load4s<TG>(dxg + rr * H + c); }
        else { xv[q][i] = make_float4(0, 0, 0, 0); dv[q][i] = make_float4(0, 0, 0, 0); }
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (q == 1 && !two) break;
      while (e + 1 < E && offsets[e + 1] <= r + q) { flush_atomic(e); flushed = true; ++e; load_gamma(e); }
      float s1 = 0.f, s2 = 0.f;
      float4 xh[IT], gd[IT];
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        int c = (lane + 64 * i) * 4;
        const float4 xq = xv[q][i], dq = dv[q][i];
        xh[i] = make_float4((xq.x - mean[q]) * rstd[q], (xq.y - mean[q]) * rstd[q], (xq.z - mean[q]) * rstd[q],
                            (xq.w - mean[q]) * rstd[q]);
        gd[i] = make_float4(dq.x * g4[i].x, dq.y * g4[i].y, dq.z * g4[i].z, dq.w * g4[i].w);
        if (c < H) {
          ag[i].x += dq.x * xh[i].x; ag[i].y += dq.y * xh[i].y; ag[i].z += dq.z * xh[i].z; ag[i].w += dq.w * xh[i].w;
          ab[i].x += dq.x; ab[i].y += dq.y; ab[i].z += dq.z; ab[i].w += dq.w;
          s1 += (gd[i].x + gd[i].y) + (gd[i].z + gd[i].w);
          s2 += (gd[i].x * xh[i].x + gd[i].y * xh[i].y) + (gd[i].z * xh[i].z + gd[i].w * xh[i].w);
        }
      }
      const float m1 = wave_sum(s1) * inv_h(H), m2 = wave_sum(s2) * inv_h(H);
      if (dxr) {   // NULL: only the affine gradients are wanted
        TG *dst = dxr + (r + q) * H;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
          int c = (lane + 64 * i) * 4;
          if (c < H)
            store4<TG>(dst + c, make_float4(rstd[q] * (gd[i].x - m1 - xh[i].x * m2), rstd[q] * (gd[i].y - m1 - xh[i].y * m2),
                                            rstd[q] * (gd[i].z - m1 - xh[i].z * m2), rstd[q] * (gd[i].w - m1 - xh[i].w * m2)));
        }
      }
    }
  }
  // end of block.  Rows are expert-sorted, so almost every block lies inside ONE expert: its four waves are
  // combined in LDS and the sums go to the block's slot of `part` (folded per expert, in block order, by
  // gather_ln_fold_k).  Only blocks that straddle an expert boundary use float atomics: a few per launch
  // instead of 2H per block on 2*E*H addresses (300 us of a 490 us kernel at 196k rows).
  if (lane == 0) s_e[wv] = flushed ? -2 : e;
  __syncthreads();
  const bool uniform = s_e[0] >= 0 && s_e[0] == s_e[1] && s_e[1] == s_e[2] && s_e[2] == s_e[3];
  const int Q = H / 4;
  if (uniform) {
    if (wv > 0) {
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        int cq = lane + 64 * i;
        if (cq < Q) { red[((wv - 1) * 2 + 0) * Q + cq] = ag[i]; red[((wv - 1) * 2 + 1) * Q + cq] = ab[i]; }
      }
    }
    __syncthreads();
    if (wv == 0) {
      float *dst = part + (int64_t)blockIdx.x * 2 * H;
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        int cq = lane + 64 * i;
        if (cq < Q) {
          for (int w = 0; w < 3; ++w) {
            float4 u = red[(w * 2 + 0) * Q + cq], v = red[(w * 2 + 1) * Q + cq];
            ag[i].x += u.x; ag[i].y += u.y; ag[i].z += u.z; ag[i].w += u.w;
            ab[i].x += v.x; ab[i].y += v.y; ab[i].z += v.z; ab[i].w += v.w;
          }
          *reinterpret_cast<float4 *>(dst + cq * 4) = ag[i];
          *reinterpret_cast<float4 *>(dst + H + cq * 4) = ab[i];
        }
      }
      if (lane == 0) blk_expert[blockIdx.x] = e;
    }
  } else {
    if (e >= 0) flush_atomic(e);
    if (threadIdx.x == 0) blk_expert[blockIdx.x] = -1;
  }
}

// dgamma[e] += sum over the blocks whose slot belongs to expert e, in block order (fixed); dbeta likewise.
// grid = (ceil(2H/64), E), 1024 threads
__global__ void __launch_bounds__(1024)
gather_ln_fold_k(const float *__restrict__ part, const int32_t *__restrict__ blk_expert, const int32_t *__restrict__ offsets,
                 float *__restrict__ dgamma, float *__restrict__ dbeta, int64_t max_rows, int64_t nblk, int H, int E) {
  __shared__ float red[16][64];
  const int e = blockIdx.y, lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int64_t total = min((int64_t)offsets[E], max_rows);
  const int64_t ra = min((int64_t)offsets[e], total), rb = min((int64_t)offsets[e + 1], total);
  float s = 0.f;
  if (c < 2 * H && rb > ra) {
    const int64_t b0 = ra / (4 * GLN_RPW), b1 = min((rb - 1) / (4 * GLN_RPW), nblk - 1);
    for (int64_t b = b0 + seg; b <= b1; b += 16)
      if (blk_expert[b] == e) s += part[b * 2 * H + c];
  }
  red[seg][lane] = s;
  __syncthreads();
  if (seg == 0 && c < 2 * H) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i][lane];
    float *dst = c < H ? dgamma + (int64_t)e * H + c : dbeta + (int64_t)e * H + (c - H);
    *dst += t;   // on top of the boundary blocks' atomics (this kernel runs after them, one writer per element)
  }
}

// out[s,:] = sum_k (w[s,k] or 1) * yr[slot_of[s,k],:], k ascending (== index_add_ order, core.py:605)
template <typename TY, typename TO, int IT>
__global__ void __launch_bounds__(256)
combine_fwd_k(const TY *__restrict__ yr, const int32_t *__restrict__ slot_of, const float *__restrict__ wk,
              TO *__restrict__ out, int64_t S, int H, int K, int with_w) {
  if constexpr (IT <= 4) __builtin_assume(H > 256 * (IT - 1));   // IT = ceil(H / 256): only the last chunk needs its bounds test
  const int lane = threadIdx.x & 63;
  const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= S) return;
  float4 acc[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) acc[i] = make_float4(0, 0, 0, 0);
  for (int k = 0; k < K; ++k) {
    const int slot = slot_of[s * K + k];
    if (slot < 0) continue;
    const float wv = with_w ? wk[s * K + k] : 1.f;
    const TY *src = yr + (int64_t)slot * H;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int c = (lane + 64 * i) * 4;
      if (c < H) {
        float4 v = load4s<TY>(src + c);
        // separate multiply then add, like `expert_output * weights` followed by index_add_
        acc[i].x += v.x * wv; acc[i].y += v.y * wv; acc[i].z += v.z * wv; acc[i].w += v.w * wv;
      }
    }
  }
  TO *dst = out + s * H;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    int c = (lane + 64 * i) * 4;
    if (c < H) store4<TO>(dst + c, acc[i]);
  }
}

// dyr[r,:] = w[s,k]*dout[s,:];  dwk[s,k] = <dout[s,:], yr[r,:]>
template <typename TD, typename TY, int IT>
__global__ void __launch_bounds__(256)
combine_bwd_k(const TD *__restrict__ dout, const TY *__restrict__ yr, const int32_t *__restrict__ row_token,
              const int32_t *__restrict__ row_k, const int32_t *__restrict__ offsets,
              const float *__restrict__ wk, TY *__restrict__ dyr, float *__restrict__ dwk, int64_t max_rows,
              int H, int K, int E) {
  if constexpr (IT <= 4) __builtin_assume(H > 256 * (IT - 1));   // IT = ceil(H / 256): only the last chunk needs its bounds test
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= max_rows || r >= offsets[E]) return;
  const int64_t s = row_token[r];
  const int k = row_k[r];
  const float wv = wk[s * K + k];
  const TD *dsrc = dout + s * H;
  const TY *ysrc = yr + r * H;
  TY *dst = dyr + r * H;
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    int c = (lane + 64 * i) * 4;
    if (c < H) {
      float4 d = load4<TD>(dsrc + c), y = load4s<TY>(ysrc + c);
      dot += (d.x * y.x + d.y * y.y) + (d.z * y.z + d.w * y.w);
      store4<TY>(dst + c, make_float4(d.x * wv, d.y * wv, d.z * wv, d.w * wv));
    }
  }
  dot = wave_sum(dot);
  if (lane == 0) dwk[s * K + k] = dot;
}

// Plain LayerNorm backward.  Block = 4 waves x LN_RPW rows each, two rows in flight per wave
// (the row loop is latency-bound otherwise); dx is written in x's dtype (the fp32 residual
// stream); the affine gradients are reduced over the block's waves in LDS and leave as ONE
// partial row per block, folded in a fixed order by ln_fold_k (deterministic, no atomics).
#ifndef APERTIS_LN_RPW
#define APERTIS_LN_RPW 8
#endif
constexpr int LN_RPW = APERTIS_LN_RPW;

// Block boundary of the pre-norm stack, forward: y = res + dropout(blk) (the residual stream, core.py:698,888)
// and xn = LayerNorm(y) (the next sub-block's pre-norm, core.py:667,847) in one pass: as two kernels y is
// written by the first and read back by the second (T*H*4 bytes each way).  Wave per row; the mask is the
// counter hash of (seed, linear index) that apertis_dropout_add_fwd uses, so the backward regenerates it.
template <typename TX, typename TO, int IT>
__global__ void __launch_bounds__(256)
dropadd_ln_fwd_k(const TO *__restrict__ blk, const int32_t *__restrict__ slot_of, const float *__restrict__ wk, int K,
                 const TX *__restrict__ res, const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                 TX *__restrict__ y, TO *__restrict__ xn, float *__restrict__ mean_o, float *__restrict__ rstd_o, int64_t T,
                 int H, float drop_p, uint64_t seed) {
  // the dispatch picks IT = ceil(H / 256) for IT <= 4: every chunk below the last lies inside the row, no bounds test needed
  if constexpr (IT <= 4) __builtin_assume(H > 256 * (IT - 1));
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= T) return;
  const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const uint32_t th = (uint32_t)(drop_p * 65536.f);
  float4 v[IT];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < H) {
      float4 a;
      if (slot_of) {
        // blk is the expert output [rows,H]: the token's row is the MoE combine (apertis_moe_combine_fwd: k
        // ascending, multiply then add, rounded to the block dtype) computed here instead of in a pass of its own
        float4 acc = make_float4(0, 0, 0, 0);
        for (int k = 0; k < K; ++k) {
          const int slot = slot_of[r * K + k];
          if (slot < 0) continue;
          const float wv = wk[r * K + k];
          const float4 v = load4s<TO>(blk + (int64_t)slot * H + c);
          acc.x += v.x * wv; acc.y += v.y * wv; acc.z += v.z * wv; acc.w += v.w * wv;
        }
        a = make_float4(to_f32(from_f32<TO>(acc.x)), to_f32(from_f32<TO>(acc.y)), to_f32(from_f32<TO>(acc.z)), to_f32(from_f32<TO>(acc.w)));
      } else {
        a = load4s<TO>(blk + r * H + c);
      }
      const float4 rr = load4s<TX>(res + r * H + c);
      float e[4] = {a.x, a.y, a.z, a.w};
      if (drop_p > 0.f) {
        bool keep[4];
        drop_keep4(seed, (uint64_t)r * (uint64_t)H + (uint64_t)c, th, keep);
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = keep[j] ? e[j] * ks : 0.f;
      }
      v[i] = make_float4(rr.x + e[0], rr.y + e[1], rr.z + e[2], rr.w + e[3]);
      store4<TX>(y + r * H + c, v[i]);
      // the norm sees y as stored (a no-op for the fp32 stream)
      v[i] = make_float4(to_f32(from_f32<TX>(v[i].x)), to_f32(from_f32<TX>(v[i].y)), to_f32(from_f32<TX>(v[i].z)), to_f32(from_f32<TX>(v[i].w)));
      sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    } else {
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const float mean = wave_sum(sum) * inv_h(H);
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < H) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      sq += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) * inv_h(H) + eps);
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < H) {
      const float4 g4 = load4<float>(gamma + c), b4 = load4<float>(beta + c);
      store4<TO>(xn + r * H + c, make_float4((v[i].x - mean) * rstd * g4.x + b4.x, (v[i].y - mean) * rstd * g4.y + b4.y,
                                             (v[i].z - mean) * rstd * g4.z + b4.z, (v[i].w - mean) * rstd * g4.w + b4.w));
    }
  }
  if (lane == 0) { mean_o[r] = mean; rstd_o[r] = rstd; }
}

// COMB (round 6): the block output was the MoE combine of expert rows (dropadd_ln_fwd_k with slot_of): the masked gradient row
// `dblk` is not stored - the row's combine backward (combine_bwd_k: dyr[slot] = w * dblk, dwk = <dblk, yr[slot]>, on dblk as
// that kernel would have read it back, rounded to TG) runs here on the row in registers, for its K <= 2 slots.  Saves the
// write and the read of [T, H] in the compute dtype (2 x 254 MB per layer at the bench shape) and a launch; same arithmetic
// per output as the two kernels: dyr and dwk bit-identical.
constexpr int LN_COMB_K = 2;
// waves per SIMD the COMB form is compiled for at H = 513..768 (IT = 3): 3 = the plain form's occupancy (168 registers, a few
// spilled: 482-500 us at the bench shape, alone), 1 = what the compiler takes by itself (two waves per SIMD, both rows' expert
// rows in flight with the rows: 461-465 us)
#ifndef APERTIS_LN_COMB_WAVES
#define APERTIS_LN_COMB_WAVES 1
#endif
constexpr int LN_COMB_WAVES = APERTIS_LN_COMB_WAVES;
template <typename TX, typename TG, int IT, bool COMB = false>
__global__ void __launch_bounds__(256, COMB && IT == 3 ? LN_COMB_WAVES : 1)
layernorm_bwd_k(const TX *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ mean_i,
                const float *__restrict__ rstd_i, const TG *__restrict__ dy, const TX *__restrict__ dres,
                TX *__restrict__ dx, TG *__restrict__ dblk, float drop_p, uint64_t seed, float *__restrict__ part,
                int64_t T, int H, const int32_t *__restrict__ slot_of = nullptr, const float *__restrict__ wk = nullptr,
                int K = 0, const TG *__restrict__ yr = nullptr, TG *__restrict__ dyr = nullptr,
                float *__restrict__ dwk = nullptr) {
  typedef typename raw4<TG>::type rawg_t;
  // the dispatch picks IT = ceil(H / 256) for IT <= 4: every chunk below the last lies inside the row, no bounds test needed
  if constexpr (IT <= 4) __builtin_assume(H > 256 * (IT - 1));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4 *red = reinterpret_cast<float4 *>(smem);  // [3 waves][2][H/4]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + wv) * LN_RPW, r1 = min(r0 + LN_RPW, T);
  float4 ag[IT], ab[IT], g4[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    ag[i] = make_float4(0, 0, 0, 0); ab[i] = make_float4(0, 0, 0, 0);
    int c = (lane + 64 * i) * 4;
    g4[i] = c < H ? load4<float>(gamma + c) : make_float4(0, 0, 0, 0);
  }
  for (int64_t r = r0; r < r1; r += 2) {
    const bool two = r + 1 < r1;
    // (COMB: the incoming gradient rows wait in their storage form - half the registers for bf16 - to make room for the expert rows)
    typedef typename std::conditional<COMB, rawg_t, float4>::type dv_t;
    float4 xv[2][IT];
    dv_t dv[2][IT];
    [[maybe_unused]] int slot[2][LN_COMB_K];
    [[maybe_unused]] float wsl[2][LN_COMB_K];
    [[maybe_unused]] rawg_t yv[2][LN_COMB_K][IT];
    float mean[2], rstd[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int64_t rr = (q == 0 || two) ? r + q : r;
      mean[q] = mean_i[rr]; rstd[q] = rstd_i[rr];
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        int c = (lane + 64 * i) * 4;
        if (c < H) {
          xv[q][i] = load4s<TX>(x + rr * H + c);
          if constexpr (COMB) dv[q][i] = raw_load(dy + rr * H + c); else dv[q][i] = load4s<TG>(dy + rr * H + c);
        } else { xv[q][i] = make_float4(0, 0, 0, 0); dv[q][i] = dv_t{}; }
      }
      if constexpr (COMB) {   // the row's expert rows: in flight with the row itself
#pragma unroll
        for (int k = 0; k < LN_COMB_K; ++k) {
          slot[q][k] = k < K ? __builtin_amdgcn_readfirstlane(slot_of[rr * K + k]) : -1;
          wsl[q][k] = slot[q][k] >= 0 ? wk[rr * K + k] : 0.f;
#pragma unroll
          for (int i = 0; i < IT; ++i) {
            const int c = (lane + 64 * i) * 4;
            yv[q][k][i] = (slot[q][k] >= 0 && c < H) ? raw_load(yr + (int64_t)slot[q][k] * H + c) : rawg_t{};
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (q == 1 && !two) break;
      [[maybe_unused]] float dot[LN_COMB_K] = {0.f, 0.f};
      float s1 = 0.f, s2 = 0.f;
      float4 xh[IT], gd[IT];
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        int c = (lane + 64 * i) * 4;
        const float4 xq = xv[q][i], dq = raw_to_f4(dv[q][i]);
        xh[i] = make_float4((xq.x - mean[q]) * rstd[q], (xq.y - mean[q]) * rstd[q], (xq.z - mean[q]) * rstd[q],
                            (xq.w - mean[q]) * rstd[q]);
        gd[i] = make_float4(dq.x * g4[i].x, dq.y * g4[i].y, dq.z * g4[i].z, dq.w * g4[i].w);
        if (c < H) {
          ag[i].x += dq.x * xh[i].x; ag[i].y += dq.y * xh[i].y; ag[i].z += dq.z * xh[i].z; ag[i].w += dq.w * xh[i].w;
          ab[i].x += dq.x; ab[i].y += dq.y; ab[i].z += dq.z; ab[i].w += dq.w;
          s1 += (gd[i].x + gd[i].y) + (gd[i].z + gd[i].w);
          s2 += (gd[i].x * xh[i].x + gd[i].y * xh[i].y) + (gd[i].z * xh[i].z + gd[i].w * xh[i].w);
        }
      }
      const float m1 = wave_sum(s1) * inv_h(H), m2 = wave_sum(s2) * inv_h(H);
      TX *dst = dx + (r + q) * H;
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        int c = (lane + 64 * i) * 4;
        if (c < H) {
          // dres: the gradient arriving on the residual branch that bypasses this norm (pre-norm block
          // y = x + f(LN(x))): added here instead of in a separate full-width pass
          const float4 rr = dres ? load4s<TX>(dres + (r + q) * H + c) : make_float4(0, 0, 0, 0);
          const float4 dt = make_float4(rstd[q] * (gd[i].x - m1 - xh[i].x * m2) + rr.x, rstd[q] * (gd[i].y - m1 - xh[i].y * m2) + rr.y,
                                        rstd[q] * (gd[i].z - m1 - xh[i].z * m2) + rr.z, rstd[q] * (gd[i].w - m1 - xh[i].w * m2) + rr.w);
          store4<TX>(dst + c, dt);
          if (COMB || dblk) {
            // block boundary, backward: x was res + dropout(blk), so the block output's gradient is the masked
            // copy of this row's total gradient (what apertis_dropout_bwd computes in a pass of its own)
            float e[4] = {dt.x, dt.y, dt.z, dt.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = to_f32(from_f32<TX>(e[j]));
            if (drop_p > 0.f) {
              bool keep[4];
              drop_keep4(seed, (uint64_t)(r + q) * (uint64_t)H + (uint64_t)c, (uint32_t)(drop_p * 65536.f), keep);
              const float ks = 1.f / (1.f - drop_p);
#pragma unroll
              for (int j = 0; j < 4; ++j) e[j] = keep[j] ? e[j] * ks : 0.f;
            }
            if constexpr (COMB) {
              // d = dblk as combine_bwd_k reads it back (rounded to TG); its arithmetic, operation for operation
              const float4 d = make_float4(to_f32(from_f32<TG>(e[0])), to_f32(from_f32<TG>(e[1])), to_f32(from_f32<TG>(e[2])),
                                           to_f32(from_f32<TG>(e[3])));
#pragma unroll
              for (int k = 0; k < LN_COMB_K; ++k) {
                if (slot[q][k] >= 0) {   // (wave-uniform)
                  const float4 y = raw_to_f4(yv[q][k][i]);
                  dot[k] += (d.x * y.x + d.y * y.y) + (d.z * y.z + d.w * y.w);
                  store4<TG>(dyr + (int64_t)slot[q][k] * H + c,
                             make_float4(d.x * wsl[q][k], d.y * wsl[q][k], d.z * wsl[q][k], d.w * wsl[q][k]));
                }
              }
            } else {
              store4<TG>(dblk + (r + q) * H + c, make_float4(e[0], e[1], e[2], e[3]));
            }
          }
        }
      }
      if constexpr (COMB) {
#pragma unroll
        for (int k = 0; k < LN_COMB_K; ++k) {
          if (slot[q][k] >= 0) {
            const float dsum = wave_sum(dot[k]);
            if (lane == 0) dwk[(r + q) * K + k] = dsum;
          }
        }
      }
    }
  }
  // block reduction: waves 1..3 park their sums in LDS, wave 0 adds them in order and writes
  const int Q = H / 4;
  if (wv > 0) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int cq = lane + 64 * i;
      if (cq < Q) { red[((wv - 1) * 2 + 0) * Q + cq] = ag[i]; red[((wv - 1) * 2 + 1) * Q + cq] = ab[i]; }
    }
  }
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int cq = lane + 64 * i;
      if (cq < Q) {
        float4 a = ag[i], b = ab[i];
        for (int w = 0; w < 3; ++w) {
          float4 u = red[(w * 2 + 0) * Q + cq], v = red[(w * 2 + 1) * Q + cq];
          a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
          b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
        }
        *reinterpret_cast<float4 *>(part + ((int64_t)blockIdx.x * 2 + 0) * H + cq * 4) = a;
        *reinterpret_cast<float4 *>(part + ((int64_t)blockIdx.x * 2 + 1) * H + cq * 4) = b;
      }
    }
  }
}

// out[c] = sum_r part[r][c] over c in [0, 2H): first H -> dgamma, next H -> dbeta (fixed order).  With `fold_out` the launch is
// the FIRST of two levels: block (x, y) sums the rows [y*rpg, (y+1)*rpg) into fold_out[y][c] (a single level leaves all of
// `part` - 32 MB per call at 180 k rows - to 2H/64 = 22 work-groups: 31 us, a tenth of the LayerNorm backward itself)
__global__ void __launch_bounds__(1024)
ln_fold_k(const float *__restrict__ part, float *__restrict__ dgamma, float *__restrict__ dbeta, int64_t nrows, int H,
          float *__restrict__ fold_out, int64_t rpg) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int64_t r0 = fold_out ? (int64_t)blockIdx.y * rpg : 0, r1 = fold_out ? min(r0 + rpg, nrows) : nrows;
  float s = 0.f;
  if (c < 2 * H) {
    int64_t w = r0 + seg;
    for (; w + 48 < r1; w += 64) {   // four independent loads in flight
      float a0 = part[w * 2 * H + c], a1 = part[(w + 16) * 2 * H + c], a2 = part[(w + 32) * 2 * H + c],
            a3 = part[(w + 48) * 2 * H + c];
      s += (a0 + a1) + (a2 + a3);
    }
    for (; w < r1; w += 16) s += part[w * 2 * H + c];
  }
  red[seg][lane] = s;
  __syncthreads();
  if (seg == 0 && c < 2 * H) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i][lane];
    if (fold_out) fold_out[(int64_t)blockIdx.y * 2 * H + c] = t;
    else if (c < H) dgamma[c] = t; else dbeta[c - H] = t;
  }
}

// ------------------------------------------------------------------------------------------
// Router projection: y[T,N] = x[T,K] W[N,K]^T + b, N <= 16 (reference core.py:430,482: H -> num_experts).
// A GEMM library spends ~115 us on this 0.4 GFLOP product (N=8); it is a bandwidth problem: one
// wave per row, the weight matrix lives in registers, N dot products are finished with wave
// reductions.  Backward: dx = dy W (row kernel), dW/db = per-wave register sums over 8 rows ->
// block partials -> fixed-order fold.
// ------------------------------------------------------------------------------------------
constexpr int SK_MAXN = 16;

template <typename TX, int IT, int NN>
__global__ void __launch_bounds__(256)
skinny_fwd_k(const TX *__restrict__ x, const float *__restrict__ W, const float *__restrict__ b, float *__restrict__ y,
             int64_t T, int K) {
  if constexpr (IT <= 4) __builtin_assume(K > 256 * (IT - 1));   // IT = ceil(K / 256): only the last chunk needs its bounds test
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  float4 w[NN][IT];
#pragma unroll
  for (int n = 0; n < NN; ++n)
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int c = (lane + 64 * i) * 4;
      w[n][i] = c < K ? load4<float>(W + (int64_t)n * K + c) : make_float4(0, 0, 0, 0);
    }
  for (int64_t r = wave; r < T; r += nw) {
    float4 xv[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int c = (lane + 64 * i) * 4;
      xv[i] = c < K ? load4s<TX>(x + r * K + c) : make_float4(0, 0, 0, 0);
    }
    float acc[NN];
#pragma unroll
    for (int n = 0; n < NN; ++n) {
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < IT; ++i) a += (xv[i].x * w[n][i].x + xv[i].y * w[n][i].y) + (xv[i].z * w[n][i].z + xv[i].w * w[n][i].w);
      acc[n] = wave_sum(a);
    }
    if (lane < NN) {
      float v = 0.f;
#pragma unroll
      for (int n = 0; n < NN; ++n) if (lane == n) v = acc[n];
      y[r * NN + lane] = v + (b ? b[lane] : 0.f);
    }
  }
}

template <typename TX, int IT, int NN>
__global__ void __launch_bounds__(256)
skinny_bwd_k(const TX *__restrict__ x, const float *__restrict__ W, const float *__restrict__ dy, TX *__restrict__ dx,
             float *__restrict__ part, int64_t T, int K) {
  if constexpr (IT <= 4) __builtin_assume(K > 256 * (IT - 1));   // IT = ceil(K / 256): only the last chunk needs its bounds test
  // part: [gridDim.x][NN*K + NN] per-block partial sums of dW (row-major [NN][K]) then db
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4 *red = reinterpret_cast<float4 *>(smem);   // [3 waves][NN][K/4]
  __shared__ float redb[4][SK_MAXN];
  constexpr int RPW = 8;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + wv) * RPW, r1 = min(r0 + RPW, T);
  float4 w[NN][IT], aw[NN][IT];
  float abias[NN];
#pragma unroll
  for (int n = 0; n < NN; ++n) {
    abias[n] = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int c = (lane + 64 * i) * 4;
      w[n][i] = c < K ? load4<float>(W + (int64_t)n * K + c) : make_float4(0, 0, 0, 0);
      aw[n][i] = make_float4(0, 0, 0, 0);
    }
  }
  for (int64_t r = r0; r < r1; ++r) {
    float g[NN];
#pragma unroll
    for (int n = 0; n < NN; ++n) g[n] = dy[r * NN + n];   // same address in every lane: one broadcast load
    float4 xv[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int c = (lane + 64 * i) * 4;
      xv[i] = c < K ? load4s<TX>(x + r * K + c) : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int c = (lane + 64 * i) * 4;
      float4 d = make_float4(0, 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NN; ++n) {
        d.x += g[n] * w[n][i].x; d.y += g[n] * w[n][i].y; d.z += g[n] * w[n][i].z; d.w += g[n] * w[n][i].w;
        aw[n][i].x += g[n] * xv[i].x; aw[n][i].y += g[n] * xv[i].y; aw[n][i].z += g[n] * xv[i].z; aw[n][i].w += g[n] * xv[i].w;
      }
      if (c < K) store4<TX>(dx + r * K + c, d);
    }
#pragma unroll
    for (int n = 0; n < NN; ++n) abias[n] += g[n];
  }
  const int Q = K / 4;
  if (wv > 0) {
#pragma unroll
    for (int n = 0; n < NN; ++n)
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        int cq = lane + 64 * i;
        if (cq < Q) red[((wv - 1) * NN + n) * Q + cq] = aw[n][i];
      }
  }
  if (lane == 0)
#pragma unroll
    for (int n = 0; n < NN; ++n) redb[wv][n] = abias[n];
  __syncthreads();
  float *dst = part + (int64_t)blockIdx.x * (NN * K + NN);
  if (wv == 0) {
#pragma unroll
    for (int n = 0; n < NN; ++n)
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        int cq = lane + 64 * i;
        if (cq < Q) {
          float4 a = aw[n][i];
          for (int w_ = 0; w_ < 3; ++w_) {
            float4 u = red[(w_ * NN + n) * Q + cq];
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
          }
          *reinterpret_cast<float4 *>(dst + (int64_t)n * K + cq * 4) = a;
        }
      }
    if (lane < NN) dst[NN * K + lane] = (redb[0][lane] + redb[1][lane]) + (redb[2][lane] + redb[3][lane]);
  }
}

// out[c] = sum_r part[r][c] for c < cols (fixed order)
__global__ void __launch_bounds__(1024)
fold_rows_k(const float *__restrict__ part, float *__restrict__ out, int64_t nrows, int64_t cols) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + lane;
  float s = 0.f;
  if (c < cols) {
    int64_t w = seg;
    for (; w + 48 < nrows; w += 64)
      s += (part[w * cols + c] + part[(w + 16) * cols + c]) + (part[(w + 32) * cols + c] + part[(w + 48) * cols + c]);
    for (; w < nrows; w += 16) s += part[w * cols + c];
  }
  red[seg][lane] = s;
  __syncthreads();
  if (seg == 0 && c < cols) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i][lane];
    out[c] = t;
  }
}

// ------------------------------------------------------------------------------------------
// gate + auxiliary losses in one pass (training): besides gates / top-K / weights, the per-row
// log-sum-exp and the three reductions the two router losses need - column sums of the gates,
// assignment counts per expert, sum of lse^2 - as per-block partials, folded in block order by
// gate_aux_fold_k, which also evaluates (reference core.py:499-505, 524-526)
//   lb = lb_coef * E * sum_e (count_e / S) * (colsum_e / S),   rz = rz_coef * sum_s lse_s^2 / S.
// As stock tensor ops the two losses are ~35 launches forward and ~25 backward per layer, 4-5 us each
// on [S, 8] tensors, plus a 47 us index_add.  stats out: [lb, rz, frac_0..frac_{E-1}].
// ------------------------------------------------------------------------------------------
// Noisy top-k routing (reference core.py:485-488): logits += randn * softplus(w_noise) * alpha.  The standard normals come
// from a counter hash of (seed, token, expert pair) through Box-Muller, so the backward regenerates them instead of keeping
// a [S, E] tensor, and the whole noise path is part of the gate kernels (as tensor ops it was ten launches per layer).
__device__ __forceinline__ void gauss_pair(uint64_t seed, uint64_t pair, float &n0, float &n1) {
  const uint32_t h0 = drop_hash_pair(seed, 2 * pair), h1 = drop_hash_pair(seed ^ 0x9E3779B97F4A7C15ull, 2 * pair + 1);
  const float u1 = ((float)(h0 >> 8) + 1.0f) * (1.0f / 16777216.0f);      // (0, 1]
  const float u2 = (float)(h1 >> 8) * (1.0f / 16777216.0f);               // [0, 1)
  const float r = sqrtf(-2.0f * logf(u1));
  float sn, cs;
  sincospif(2.0f * u2, &sn, &cs);
  n0 = r * cs;
  n1 = r * sn;
}
__device__ __forceinline__ float softplus_gate(float x) { return x > 20.f ? x : log1pf(expf(x)); }   // F.softplus (beta 1, threshold 20)
// the row's E standard normals (pairs share one Box-Muller draw; an odd last expert uses the first of its pair)
template <int CAP>
__device__ __forceinline__ void row_noise(uint64_t seed, int64_t s, int E, float (&nz)[CAP]) {
  const int64_t ppr = (E + 1) / 2;
#pragma unroll
  for (int i = 0; i < CAP; i += 2) {
    if (i < E) {
      float a, b;
      gauss_pair(seed, (uint64_t)(s * ppr + (i >> 1)), a, b);
      nz[i] = a;
      if (i + 1 < CAP) nz[i + 1] = b;
    }
  }
}

template <int EC>
__global__ void __launch_bounds__(256)
gate_topk_aux_fwd_k(const float *__restrict__ logits, const float *__restrict__ w_noise, float alpha, uint64_t seed,
                    float *__restrict__ gates, int32_t *__restrict__ idx,
                    float *__restrict__ w, float *__restrict__ lse, float *__restrict__ part, int64_t S, int E_rt, int K) {
  constexpr int CAP = EC > 0 ? EC : MAXE;
  const int E = EC > 0 ? EC : E_rt;
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = s < S;
  float v[CAP];
  float cnt[CAP];
  float l2 = 0.f;
#pragma unroll
  for (int i = 0; i < CAP; ++i) { v[i] = 0.f; cnt[i] = 0.f; }
  if (live) {
    const float *row = logits + s * E;
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < CAP; ++i)
      if (i < E) v[i] = row[i];
    if (w_noise) {
      float nz[CAP];
      row_noise<CAP>(seed, s, E, nz);
#pragma unroll
      for (int i = 0; i < CAP; ++i)
        if (i < E) v[i] += nz[i] * (softplus_gate(w_noise[i]) * alpha);
    }
#pragma unroll
    for (int i = 0; i < CAP; ++i)
      if (i < E) m = fmaxf(m, v[i]);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < CAP; ++i)
      if (i < E) { v[i] = expf(v[i] - m); sum += v[i]; }
    const float l = m + logf(sum);
    lse[s] = l;
    l2 = l * l;
#pragma unroll
    for (int i = 0; i < CAP; ++i)
      if (i < E) { v[i] = v[i] / sum; gates[s * E + i] = v[i]; }
    uint64_t chosen = 0;
    float p[MAXK];
    float psum = 0.f;
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
      if (k < K) {
        float best = -1.f;
        int bi = 0;
#pragma unroll
        for (int i = 0; i < CAP; ++i)
          if (i < E && !((chosen >> i) & 1) && v[i] > best) { best = v[i]; bi = i; }
        chosen |= 1ull << bi;
        idx[s * K + k] = bi;
        p[k] = best;
        psum += best;
      }
    }
    const float den = psum + 1e-6f;  // core.py:529
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
      if (k < K) w[s * K + k] = p[k] / den;
#pragma unroll
    for (int i = 0; i < CAP; ++i)
      if (i < E) cnt[i] = (float)((chosen >> i) & 1);
  }
  // block partials: [colsum(gates) E | counts E | sum lse^2]
  __shared__ float red[4][2 * MAXE + 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < CAP; ++i) {
    if (i < E) {
      const float a = wave_sum(v[i]), b = wave_sum(cnt[i]);
      if (lane == 0) { red[wv][i] = a; red[wv][E + i] = b; }
    }
  }
  l2 = wave_sum(l2);
  if (lane == 0) red[wv][2 * E] = l2;
  __syncthreads();
  if (threadIdx.x < 2 * E + 1)
    part[(int64_t)blockIdx.x * (2 * E + 1) + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ void __launch_bounds__(1024)
gate_aux_fold_k(const float *__restrict__ part, float *__restrict__ stats, int64_t nblk, int64_t S, int E, float lb_coef,
                float rz_coef) {
  __shared__ float tot[2 * MAXE + 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int cols = 2 * E + 1;
  for (int cc = wv; cc < cols; cc += 16) {     // a wave per column: lane-strided partial sums, then the wave (fixed order)
    float a = 0.f;
    for (int64_t b = lane; b < nblk; b += 64) a += part[b * cols + cc];
    a = wave_sum(a);
    if (lane == 0) tot[cc] = a;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float lb = 0.f;
    for (int e = 0; e < E; ++e) lb += (tot[E + e] / (float)S) * (tot[e] / (float)S);
    stats[0] = lb_coef * (float)E * lb;
    stats[1] = rz_coef * tot[2 * E] / (float)S;
  }
  if (threadIdx.x < E) stats[2 + threadIdx.x] = tot[E + threadIdx.x] / (float)S;
}

// backward of gate + losses: dgates[s,e] = dlb * lb_coef * E * frac_e / S (the load-balancing loss through the
// gate means), dlogits += drz * rz_coef * 2 lse_s / S * gates[s,e] (d lse / d logits = softmax)
template <int EC>
__global__ void __launch_bounds__(256)
gate_topk_aux_bwd_k(const float *__restrict__ gates, const int32_t *__restrict__ idx,
                    const float *__restrict__ dw, const float *__restrict__ lse,
                    const float *__restrict__ stats, const float *__restrict__ dlb,
                    const float *__restrict__ drz, float lb_coef, float rz_coef,
                    float *__restrict__ dlogits, float *__restrict__ npart, uint64_t seed, int64_t S, int E_rt, int K) {
  // npart != NULL (noisy routing): [gridDim.x][E] block sums of dlogits * n, the gradient of the per-expert noise scale
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int CAP = EC > 0 ? EC : MAXE;
  const int E = EC > 0 ? EC : E_rt;
  float dsc[CAP];
#pragma unroll
  for (int i = 0; i < CAP; ++i) dsc[i] = 0.f;
  if (s < S) {
  const float glb = dlb ? dlb[0] * lb_coef * (float)E / (float)S : 0.f;
  const float grz = drz ? drz[0] * rz_coef * 2.f * lse[s] / (float)S : 0.f;
  float g[CAP], dg[CAP];
#pragma unroll
  for (int i = 0; i < CAP; ++i)
    if (i < E) { g[i] = gates[s * E + i]; dg[i] = glb * stats[2 + i]; }
  if (dw) {
    float psum = 0.f, dot = 0.f;
    for (int k = 0; k < K; ++k) {
      int e = idx[s * K + k];
      float pe = 0.f;
#pragma unroll
      for (int i = 0; i < CAP; ++i)
        if (i == e) pe = g[i];
      psum += pe;
      dot += dw[s * K + k] * pe;
    }
    const float den = psum + 1e-6f;
    const float corr = dot / (den * den);
    for (int k = 0; k < K; ++k) {
      int e = idx[s * K + k];
      float dp = dw[s * K + k] / den - corr;
#pragma unroll
      for (int i = 0; i < CAP; ++i)
        if (i == e) dg[i] += dp;
    }
  }
  float inner = 0.f;
#pragma unroll
  for (int i = 0; i < CAP; ++i)
    if (i < E) inner += dg[i] * g[i];
  float nz[CAP];
  if (npart) row_noise<CAP>(seed, s, E, nz);
#pragma unroll
  for (int i = 0; i < CAP; ++i)
    if (i < E) {
      const float dl = g[i] * (dg[i] - inner) + grz * g[i];
      dlogits[s * E + i] = dl;
      if (npart) dsc[i] = dl * nz[i];
    }
  }
  if (npart) {
    __shared__ float red[4][MAXE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < CAP; ++i)
      if (i < E) {
        const float a = wave_sum(dsc[i]);
        if (lane == 0) red[wv][i] = a;
      }
    __syncthreads();
    if (threadIdx.x < E)
      npart[(int64_t)blockIdx.x * E + threadIdx.x] =
          (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  }
}

// d w_noise[e] = (sum over blocks of npart[., e]) * alpha * sigmoid(w_noise[e])  (softplus' = sigmoid), fixed order
__global__ void __launch_bounds__(1024)
gate_noise_fold_k(const float *__restrict__ npart, const float *__restrict__ w_noise, float alpha, float *__restrict__ dw_noise,
                  int64_t nblk, int E) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int e = wv; e < E; e += 16) {
    float a = 0.f;
    for (int64_t b = lane; b < nblk; b += 64) a += npart[b * E + e];
    a = wave_sum(a);
    if (lane == 0) {
      const float x = w_noise[e];
      dw_noise[e] = a * alpha * (x > 20.f ? 1.f : 1.f / (1.f + expf(-x)));
    }
  }
}

// ------------------------------------------------------------------------------------------
// Router projection with its LayerNorm fused in: logits = Linear(LayerNorm(x))  (reference
// core.py:481-482).  As two ops the normalised [T,H] tensor is written and read back, and the
// backward moves [T,H] five more times (skinny dx, LN dx, the add with the expert path's gradient):
// 536 us per layer at T=98304, H=704 for a layer that has 8 outputs.  Fused: the forward reads x
// once; the backward reads x and the gradient arriving on the pass-through of x (`dres`, the
// expert path) once and writes dx once.  Forward: wave per row, lanes over H as in the LayerNorm kernels.
// part: [gridDim.x][NN*H + NN + 2H] per-block sums of dW, db, dgamma, dbeta, folded in a fixed order.
// ------------------------------------------------------------------------------------------

template <typename TX, int IT, int NN>
__global__ void __launch_bounds__(256)
router_fwd_k(const TX *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
             const float *__restrict__ W, const float *__restrict__ b, float *__restrict__ logits,
             float *__restrict__ mean_o, float *__restrict__ rstd_o, int64_t T, int H) {
  // the dispatch picks IT = ceil(H / 256) for IT <= 4: every chunk below the last lies inside the row, no bounds test needed
  if constexpr (IT <= 4) __builtin_assume(H > 256 * (IT - 1));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4 *sW = reinterpret_cast<float4 *>(smem);   // [NN][H/4]: in registers the weight would cost NN*IT*4 VGPRs and two waves per SIMD
  const int lane = threadIdx.x & 63, Q = H / 4;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
  for (int i = threadIdx.x; i < NN * Q; i += 256) sW[i] = reinterpret_cast<const float4 *>(W)[i];
  float4 g4[IT], b4[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int c = (lane + 64 * i) * 4;
    g4[i] = c < H ? load4<float>(gamma + c) : make_float4(0, 0, 0, 0);
    b4[i] = c < H ? load4<float>(beta + c) : make_float4(0, 0, 0, 0);
  }
  __syncthreads();
  typedef typename raw4<TX>::type raw_t;
  raw_t cur[IT], nxt[IT];
  auto fetch = [&](raw_t (&o)[IT], int64_t r) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (lane + 64 * i) * 4;
      o[i] = (c < H && r < T) ? raw_load(x + r * H + c) : raw_t{};
    }
  };
  if (wave < T) fetch(cur, wave);
  for (int64_t r = wave; r < T; r += nw) {
    fetch(nxt, r + nw);   // the wave's next row is in flight while this one is reduced
    float4 v[IT];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      v[i] = raw_to_f4(cur[i]);
      sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(sum) * inv_h(H);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < H) {
        const float a = v[i].x - mean, bb = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
        sq += (a * a + bb * bb) + (cc * cc + d * d);
      }
    }
    const float rstd = rsqrtf(wave_sum(sq) * inv_h(H) + eps);
#pragma unroll
    for (int i = 0; i < IT; ++i) cur[i] = nxt[i];
#pragma unroll
    for (int i = 0; i < IT; ++i)   // padding lanes: g4 = b4 = 0, so xn = 0 there
      v[i] = make_float4((v[i].x - mean) * rstd * g4[i].x + b4[i].x, (v[i].y - mean) * rstd * g4[i].y + b4[i].y,
                         (v[i].z - mean) * rstd * g4[i].z + b4[i].z, (v[i].w - mean) * rstd * g4[i].w + b4[i].w);
    float acc[NN];
#pragma unroll
    for (int n = 0; n < NN; ++n) {
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const float4 wn = lane + 64 * i < Q ? sW[n * Q + lane + 64 * i] : make_float4(0, 0, 0, 0);
        a += (v[i].x * wn.x + v[i].y * wn.y) + (v[i].z * wn.z + v[i].w * wn.w);
      }
      acc[n] = wave_sum(a);
    }
    if (lane < NN) {
      float o = 0.f;
#pragma unroll
      for (int n = 0; n < NN; ++n) if (lane == n) o = acc[n];
      logits[r * NN + lane] = o + (b ? b[lane] : 0.f);
    }
    if (lane == 0) { mean_o[r] = mean; rstd_o[r] = rstd; }
  }
}

// Block boundary in front of an MoE feed-forward, forward: y = res + dropout(blk), xn = LayerNorm(y) (dropadd_ln_fwd_k) AND the
// router's logits = Linear(router_norm(xn)) (router_fwd_k) in ONE pass: xn is in registers when the boundary has normalised
// the row, so the router costs no second read of it (231 MB and a 164 us kernel per layer at the bench shape).  Same
// arithmetic, in the same order, as the two kernels it replaces - the router reads xn as stored (rounded to TO).
// Rows are walked by persistent waves (the next row's blk / res in flight), W and the two norms' affine vectors in LDS.
template <typename TX, typename TO, int IT, int NN>
__global__ void __launch_bounds__(256)
dropadd_ln_router_fwd_k(const TO *__restrict__ blk, const TX *__restrict__ res, const float *__restrict__ gamma,
                        const float *__restrict__ beta, float eps, TX *__restrict__ y, TO *__restrict__ xn,
                        float *__restrict__ mean_o, float *__restrict__ rstd_o, const float *__restrict__ rgamma,
                        const float *__restrict__ rbeta, float reps, const float *__restrict__ W, const float *__restrict__ rb,
                        float *__restrict__ logits, float *__restrict__ rmean_o, float *__restrict__ rrstd_o, int64_t T, int H,
                        float drop_p, uint64_t seed) {
  // the dispatch picks IT = ceil(H / 256) for IT <= 4: every chunk below the last lies inside the row, no bounds test needed
  if constexpr (IT <= 4) __builtin_assume(H > 256 * (IT - 1));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Q = H / 4;
  float4 *sW = reinterpret_cast<float4 *>(smem);   // [NN][Q]
  float4 *sG = sW + NN * Q, *sB = sG + Q, *sRG = sB + Q, *sRB = sRG + Q;
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < NN * Q; i += 256) sW[i] = reinterpret_cast<const float4 *>(W)[i];
  for (int i = threadIdx.x; i < Q; i += 256) {
    sG[i] = reinterpret_cast<const float4 *>(gamma)[i]; sB[i] = reinterpret_cast<const float4 *>(beta)[i];
    sRG[i] = reinterpret_cast<const float4 *>(rgamma)[i]; sRB[i] = reinterpret_cast<const float4 *>(rbeta)[i];
  }
  __syncthreads();
  const float ks = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const uint32_t th = (uint32_t)(drop_p * 65536.f);
  typedef typename raw4<TO>::type rawo_t;
  typedef typename raw4<TX>::type rawx_t;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
  rawo_t bc[IT], bn[IT];
  rawx_t rc[IT], rn[IT];
  auto fetch = [&](rawo_t (&bo)[IT], rawx_t (&ro)[IT], int64_t r) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (lane + 64 * i) * 4;
      const bool ok = c < H && r < T;
      bo[i] = ok ? *reinterpret_cast<const rawo_t *>(blk + r * H + c) : rawo_t{};
      ro[i] = ok ? *reinterpret_cast<const rawx_t *>(res + r * H + c) : rawx_t{};
    }
  };
  if (wave < T) fetch(bc, rc, wave);
  for (int64_t r = wave; r < T; r += nw) {
    fetch(bn, rn, r + nw);
    // ---- boundary: y = res + dropout(blk); statistics of y as stored
    float4 v[IT];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < H) {
        const float4 a = raw_to_f4(bc[i]), rr = raw_to_f4(rc[i]);
        float e[4] = {a.x, a.y, a.z, a.w};
        if (drop_p > 0.f) {
          bool keep[4];
          drop_keep4(seed, (uint64_t)r * (uint64_t)H + (uint64_t)c, th, keep);
#pragma unroll
          for (int j = 0; j < 4; ++j) e[j] = keep[j] ? e[j] * ks : 0.f;
        }
        v[i] = make_float4(rr.x + e[0], rr.y + e[1], rr.z + e[2], rr.w + e[3]);
        store4<TX>(y + r * H + c, v[i]);
        v[i] = make_float4(to_f32(from_f32<TX>(v[i].x)), to_f32(from_f32<TX>(v[i].y)), to_f32(from_f32<TX>(v[i].z)), to_f32(from_f32<TX>(v[i].w)));
        sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      } else {
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    const float mean = wave_sum(sum) * inv_h(H);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < H) {
        const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
        sq += (a * a + b * b) + (cc * cc + d * d);
      }
    }
    const float rstd = rsqrtf(wave_sum(sq) * inv_h(H) + eps);
    // ---- xn = LayerNorm(y), stored; the router continues on xn AS STORED
    float rsum = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < H) {
        const float4 g4 = sG[lane + 64 * i], b4 = sB[lane + 64 * i];
        const float4 o = make_float4((v[i].x - mean) * rstd * g4.x + b4.x, (v[i].y - mean) * rstd * g4.y + b4.y,
                                     (v[i].z - mean) * rstd * g4.z + b4.z, (v[i].w - mean) * rstd * g4.w + b4.w);
        store4<TO>(xn + r * H + c, o);
        v[i] = make_float4(to_f32(from_f32<TO>(o.x)), to_f32(from_f32<TO>(o.y)), to_f32(from_f32<TO>(o.z)), to_f32(from_f32<TO>(o.w)));
        rsum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      }
    }
    if (lane == 0) { mean_o[r] = mean; rstd_o[r] = rstd; }
    // ---- router: logits = Linear(router_norm(xn))  (router_fwd_k)
    const float rmean = wave_sum(rsum) * inv_h(H);
    float rsq = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < H) {
        const float a = v[i].x - rmean, b = v[i].y - rmean, cc = v[i].z - rmean, d = v[i].w - rmean;
        rsq += (a * a + b * b) + (cc * cc + d * d);
      }
    }
    const float rrstd = rsqrtf(wave_sum(rsq) * inv_h(H) + reps);
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const bool in = lane + 64 * i < Q;
      const float4 g4 = in ? sRG[lane + 64 * i] : make_float4(0, 0, 0, 0), b4 = in ? sRB[lane + 64 * i] : make_float4(0, 0, 0, 0);
      v[i] = make_float4((v[i].x - rmean) * rrstd * g4.x + b4.x, (v[i].y - rmean) * rrstd * g4.y + b4.y,
                         (v[i].z - rmean) * rrstd * g4.z + b4.z, (v[i].w - rmean) * rrstd * g4.w + b4.w);
    }
    float acc[NN];
#pragma unroll
    for (int n = 0; n < NN; ++n) {
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const float4 wn = lane + 64 * i < Q ? sW[n * Q + lane + 64 * i] : make_float4(0, 0, 0, 0);
        a += (v[i].x * wn.x + v[i].y * wn.y) + (v[i].z * wn.z + v[i].w * wn.w);
      }
      acc[n] = wave_sum(a);
    }
    if (lane < NN) {
      float o = 0.f;
#pragma unroll
      for (int n = 0; n < NN; ++n) if (lane == n) o = acc[n];
      logits[r * NN + lane] = o + (rb ? rb[lane] : 0.f);
    }
    if (lane == 0) { rmean_o[r] = rmean; rrstd_o[r] = rrstd; }
#pragma unroll
    for (int i = 0; i < IT; ++i) { bc[i] = bn[i]; rc[i] = rn[i]; }
  }
}

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// Backward, one wave per row (every lane busy, per-row scalar work done once): W in LDS, the dW / dgamma /
// dbeta accumulators in registers (2 waves per SIMD), the wave's next row prefetched while this one is
// computed.  (A form with the columns split over the block's waves - small accumulators, more waves - ran
// twice the instructions per row and was no faster: 304 vs 276 us at 98k rows.)
// MODE 0: everything in one pass.  MODE 1: dx, dgamma, dbeta only; MODE 2: dW, db only - as two launches the first runs
// without the NN * IT * 4 weight-gradient accumulators (96 of its registers at NN = 8, IT = 3) and fits two waves per SIMD,
// the second is a short kernel; both read x.  Same arithmetic per output either way.
template <typename TX, int IT, int NN, int MODE>
__global__ void __launch_bounds__(256, MODE == 1 ? 2 : 1)
router_bwd3_k(const TX *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta,
              const float *__restrict__ mean_i, const float *__restrict__ rstd_i, const float *__restrict__ W,
              const float *__restrict__ dlogits, const TX *__restrict__ dres, const TX *__restrict__ grows,
              const int32_t *__restrict__ slot_of, int KS, TX *__restrict__ dx,
              float *__restrict__ part, int64_t T, int H) {
  // the dispatch picks IT = ceil(H / 256) for IT <= 4: every chunk below the last lies inside the row, no bounds test needed
  if constexpr (IT <= 4) __builtin_assume(H > 256 * (IT - 1));
  // grows / slot_of (KS <= 2 slots per row): the gradient reaching x through the expert path as the ROWS the gather-LN
  // backward wrote - row r receives round_TX(sum_k grows[slot_of[r, k]]), k ascending, exactly what apertis_moe_combine_fwd
  // would have written into a dense `dres` (231 MB written and read back per layer at the bench shape, and a 128 us kernel)
  typedef typename raw4<TX>::type raw_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4 *sW = reinterpret_cast<float4 *>(smem);                 // [NN][H/4]
  float4 *red = sW + NN * (H / 4);                               // [NN + 2][H/4], one wave at a time
  __shared__ float redb[4][SK_MAXN];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, Q = H / 4;
  for (int i = threadIdx.x; i < NN * Q; i += 256) sW[i] = reinterpret_cast<const float4 *>(W)[i];
  float4 aw[NN][IT], ag[IT], ab[IT], g4[IT], b4[IT];
  float abias = 0.f;   // lane n < NN accumulates db[n]
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int c = (lane + 64 * i) * 4;
    g4[i] = c < H ? load4<float>(gamma + c) : make_float4(0, 0, 0, 0);
    b4[i] = c < H ? load4<float>(beta + c) : make_float4(0, 0, 0, 0);
    ag[i] = make_float4(0, 0, 0, 0); ab[i] = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int n = 0; n < NN; ++n) aw[n][i] = make_float4(0, 0, 0, 0);
  }
  __syncthreads();
  const int64_t wave = (int64_t)blockIdx.x * 4 + wv, nw = (int64_t)gridDim.x * 4;
  raw_t xc[IT], rc[IT], xn_[IT], rn_[IT];
  raw_t gc[2][IT], gn_[2][IT];       // the (up to two) gathered gradient rows of the current / next row
  // their slots ride in ONE register per row like the row's scalars below (lane k < KS holds slot_of[r, k]), fetched TWO rows
  // ahead of their use as addresses, read back with v_readlane
  int sc[2] = {-1, -1}, sn[2] = {-1, -1};
  const bool gath = MODE != 2 && grows != nullptr;
  auto fetch_slots = [&](int64_t r) -> int { return (gath && r < T && lane < KS) ? slot_of[r * KS + lane] : -1; };
  auto slots_of = [&](int v, int (&so)[2]) {
    so[0] = __builtin_amdgcn_readlane(v, 0);
    so[1] = __builtin_amdgcn_readlane(v, 1);
  };
  int slv_n = -1, slv_n2 = -1;
  auto fetch = [&](raw_t (&xo)[IT], raw_t (&ro)[IT], raw_t (&go)[2][IT], const int (&so)[2], int64_t r) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (lane + 64 * i) * 4;
      const bool ok = c < H && r < T;
      xo[i] = ok ? raw_load(x + r * H + c) : raw_t{};
      ro[i] = (MODE != 2 && ok && dres) ? raw_load(dres + r * H + c) : raw_t{};
#pragma unroll
      for (int k = 0; k < 2; ++k)
        go[k][i] = (gath && ok && so[k] >= 0) ? raw_load(grows + (int64_t)so[k] * H + c) : raw_t{};
    }
  };
  // the row's scalars - NN logit gradients, mean, rstd - ride in ONE register: lane n < NN holds dlogits[r][n],
  // lanes NN / NN+1 hold mean / rstd; fetched a row ahead like x, read back with v_readlane.  (As per-row
  // broadcast loads they were ten dependent memory round trips per row and bounded the kernel.)
  auto fetch_meta = [&](int64_t r) -> float {
    if (r >= T) return 0.f;
    const float *p = lane < NN ? dlogits + r * NN + lane : (lane == NN ? mean_i + r : rstd_i + r);
    return lane < NN + 2 ? *p : 0.f;
  };
  auto lane_val = [](float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); };
  float meta = 0.f, meta_next = 0.f;
  if (wave < T) {
    slots_of(fetch_slots(wave), sc);
    slv_n = fetch_slots(wave + nw);
    fetch(xc, rc, gc, sc, wave);
    meta = fetch_meta(wave);
  }
  for (int64_t r = wave; r < T; r += nw) {
    slv_n2 = fetch_slots(r + 2 * nw);
    slots_of(slv_n, sn);
    fetch(xn_, rn_, gn_, sn, r + nw);
    meta_next = fetch_meta(r + nw);
    float g[NN];
#pragma unroll
    for (int n = 0; n < NN; ++n) g[n] = lane_val(meta, n);
    if (MODE != 1 && lane < NN) abias += meta;
    const float mean = lane_val(meta, NN), rstd = lane_val(meta, NN + 1);
    float4 xh[IT], dn[IT];
    float s1 = 0.f, s2 = 0.f;
    // the row's arithmetic runs on the packed fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two floats per
    // lane per instruction, the same IEEE fma per element): the accumulators keep the kernel at one wave per SIMD, where
    // time follows the instruction count (PMC: VALU busy a third of the time, 494 VALU instructions per row before)
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const bool in = lane + 64 * i < Q;
      const float4 xv = raw_to_f4(xc[i]);
      const v2f mm = {-mean, -mean}, rs = {rstd, rstd};
      v2f xh0 = in ? ((v2f){xv.x, xv.y} + mm) * rs : (v2f){0.f, 0.f}, xh1 = in ? ((v2f){xv.z, xv.w} + mm) * rs : (v2f){0.f, 0.f};
      xh[i] = make_float4(xh0.x, xh0.y, xh1.x, xh1.y);
      const v2f g0 = {g4[i].x, g4[i].y}, g1 = {g4[i].z, g4[i].w};
      const v2f xn0 = pk_fma2(xh0, g0, (v2f){b4[i].x, b4[i].y}), xn1 = pk_fma2(xh1, g1, (v2f){b4[i].z, b4[i].w});
      v2f d0 = {0.f, 0.f}, d1 = {0.f, 0.f};   // dxn = dlogits @ W
#pragma unroll
      for (int n = 0; n < NN; ++n) {
        const float4 wn = in ? sW[n * Q + lane + 64 * i] : make_float4(0, 0, 0, 0);
        const v2f gn = {g[n], g[n]};
        if constexpr (MODE != 2) { d0 = pk_fma2(gn, (v2f){wn.x, wn.y}, d0); d1 = pk_fma2(gn, (v2f){wn.z, wn.w}, d1); }
        if constexpr (MODE != 1) {
          v2f a0 = {aw[n][i].x, aw[n][i].y}, a1 = {aw[n][i].z, aw[n][i].w};
          a0 = pk_fma2(gn, xn0, a0); a1 = pk_fma2(gn, xn1, a1);
          aw[n][i] = make_float4(a0.x, a0.y, a1.x, a1.y);
        }
      }
      if constexpr (MODE == 2) continue;
      v2f ag0 = pk_fma2(d0, xh0, (v2f){ag[i].x, ag[i].y}), ag1 = pk_fma2(d1, xh1, (v2f){ag[i].z, ag[i].w});
      ag[i] = make_float4(ag0.x, ag0.y, ag1.x, ag1.y);
      const v2f ab0 = (v2f){ab[i].x, ab[i].y} + d0, ab1 = (v2f){ab[i].z, ab[i].w} + d1;
      ab[i] = make_float4(ab0.x, ab0.y, ab1.x, ab1.y);
      const v2f dn0 = d0 * g0, dn1 = d1 * g1;
      dn[i] = make_float4(dn0.x, dn0.y, dn1.x, dn1.y);
      s1 += (dn[i].x + dn[i].y) + (dn[i].z + dn[i].w);
      s2 += (dn[i].x * xh[i].x + dn[i].y * xh[i].y) + (dn[i].z * xh[i].z + dn[i].w * xh[i].w);
    }
    if constexpr (MODE != 2) {
    const float m1 = wave_sum(s1) * inv_h(H), m2 = wave_sum(s2) * inv_h(H);
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < H) {
        float4 rr = raw_to_f4(rc[i]);
        if (gath) {
          float4 ga = make_float4(0, 0, 0, 0);
#pragma unroll
          for (int k = 0; k < 2; ++k)
            if (sc[k] >= 0) { const float4 gv = raw_to_f4(gc[k][i]); ga.x += gv.x; ga.y += gv.y; ga.z += gv.z; ga.w += gv.w; }
          rr.x += to_f32(from_f32<TX>(ga.x)); rr.y += to_f32(from_f32<TX>(ga.y));
          rr.z += to_f32(from_f32<TX>(ga.z)); rr.w += to_f32(from_f32<TX>(ga.w));
        }
        store4<TX>(dx + r * H + c, make_float4(rstd * (dn[i].x - m1 - xh[i].x * m2) + rr.x, rstd * (dn[i].y - m1 - xh[i].y * m2) + rr.y,
                                                rstd * (dn[i].z - m1 - xh[i].z * m2) + rr.z, rstd * (dn[i].w - m1 - xh[i].w * m2) + rr.w));
      }
    }
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) { xc[i] = xn_[i]; rc[i] = rn_[i]; gc[0][i] = gn_[0][i]; gc[1][i] = gn_[1][i]; }
    sc[0] = sn[0]; sc[1] = sn[1];
    slv_n = slv_n2;
    meta = meta_next;
  }
  // block reduction in wave order (waves 1..3 take turns in one LDS buffer), then one partial row per block
  if (MODE != 1 && lane < NN) redb[wv][lane] = abias;
  for (int turn = 1; turn < 4; ++turn) {
    __syncthreads();
    if (wv == turn) {
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int cq = lane + 64 * i;
        if (cq < Q) {
          if constexpr (MODE != 1) {
#pragma unroll
            for (int n = 0; n < NN; ++n) red[n * Q + cq] = aw[n][i];
          }
          if constexpr (MODE != 2) { red[NN * Q + cq] = ag[i]; red[(NN + 1) * Q + cq] = ab[i]; }
        }
      }
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int cq = lane + 64 * i;
        if (cq < Q) {
          if constexpr (MODE != 1) {
#pragma unroll
            for (int n = 0; n < NN; ++n) {
              const float4 u = red[n * Q + cq];
              aw[n][i].x += u.x; aw[n][i].y += u.y; aw[n][i].z += u.z; aw[n][i].w += u.w;
            }
          }
          if constexpr (MODE != 2) {
            const float4 u = red[NN * Q + cq], v = red[(NN + 1) * Q + cq];
            ag[i].x += u.x; ag[i].y += u.y; ag[i].z += u.z; ag[i].w += u.w;
            ab[i].x += v.x; ab[i].y += v.y; ab[i].z += v.z; ab[i].w += v.w;
          }
        }
      }
    }
  }
  float *dst = part + (int64_t)blockIdx.x * (NN * H + NN + 2 * H);
  if (wv == 0) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int cq = lane + 64 * i;
      if (cq < Q) {
        if constexpr (MODE != 1) {
#pragma unroll
          for (int n = 0; n < NN; ++n) *reinterpret_cast<float4 *>(dst + (int64_t)n * H + cq * 4) = aw[n][i];
        }
        if constexpr (MODE != 2) {
          *reinterpret_cast<float4 *>(dst + NN * H + NN + cq * 4) = ag[i];
          *reinterpret_cast<float4 *>(dst + NN * H + NN + H + cq * 4) = ab[i];
        }
      }
    }
    if (MODE != 1 && lane < NN) dst[NN * H + lane] = (redb[0][lane] + redb[1][lane]) + (redb[2][lane] + redb[3][lane]);
  }
}

// ------------------------------------------------------------------------------------------
// The entrance of an MoE feed-forward for a handful of rows (the single-token decode step, core.py:1578-1603; S <= 16), ONE
// launch of one work-group: the block boundary y = res + blk with xn = LayerNorm(y), the router's norm + projection on xn
// (dropadd_ln_router_fwd_k's row arithmetic, a wave per row; no dropout: inference), then gate, dispatch plan and the
// gather-LayerNorm of moe_route_small_k with xn taken from LDS instead of from HBM.  As separate launches these were two
// dependent 5-8 us kernels per layer of a token step.  y [S,H] TX, logits / gates / idx / w, the plan, xg [S*K,H] TO come
// out; xn only if the caller wants it (xn_o != NULL).
// ------------------------------------------------------------------------------------------
template <typename TX, typename TO, int IT, int NN>
__global__ void __launch_bounds__(1024)
moe_enter_small_k(const TO *__restrict__ blk, const TX *__restrict__ res, const float *__restrict__ gamma,
                  const float *__restrict__ beta, float eps, TX *__restrict__ y, TO *__restrict__ xn_o,
                  const float *__restrict__ rgamma, const float *__restrict__ rbeta, float reps, const float *__restrict__ W,
                  const float *__restrict__ rb, float *__restrict__ logits, float *__restrict__ gates, int32_t *__restrict__ idx_o,
                  float *__restrict__ w_o, int32_t *__restrict__ offsets, int32_t *__restrict__ row_token,
                  int32_t *__restrict__ row_k, int32_t *__restrict__ slot_of, const float *__restrict__ lgamma,
                  const float *__restrict__ lbeta, float leps, TO *__restrict__ xg, float *__restrict__ mean_o,
                  float *__restrict__ rstd_o, int S, int K, int H) {
  if constexpr (IT <= 4) __builtin_assume(H > 256 * (IT - 1));
  typedef typename raw4<TO>::type rawo_t;
  typedef typename raw4<TX>::type rawx_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Q = H / 4;
  float4 *sW = reinterpret_cast<float4 *>(smem);   // [NN][Q]
  float4 *sG = sW + NN * Q, *sB = sG + Q, *sRG = sB + Q, *sRB = sRG + Q;
  float4 *sLG = sRB + Q, *sLB = sLG + NN * Q;         // [NN][Q] each: the experts' LayerNorm vectors (the last phase meets them
                                                      // with the expert known only then: fetched from HBM there they were a round trip in the kernel's tail)
  rawo_t *sXN = reinterpret_cast<rawo_t *>(sLB + NN * Q);   // [S][Q]: xn as stored
  __shared__ int32_t s_idx[16 * MAXK], s_off[17], s_rtok[16 * MAXK];
  __shared__ float s_w[16 * MAXK], s_lg[16 * NN];
  const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6, nwaves = (int)blockDim.x >> 6;
  for (int i = t; i < NN * Q; i += (int)blockDim.x) {
    sW[i] = reinterpret_cast<const float4 *>(W)[i];
    sLG[i] = reinterpret_cast<const float4 *>(lgamma)[i];
    sLB[i] = reinterpret_cast<const float4 *>(lbeta)[i];
  }
  for (int i = t; i < Q; i += (int)blockDim.x) {
    sG[i] = reinterpret_cast<const float4 *>(gamma)[i]; sB[i] = reinterpret_cast<const float4 *>(beta)[i];
    sRG[i] = reinterpret_cast<const float4 *>(rgamma)[i]; sRB[i] = reinterpret_cast<const float4 *>(rbeta)[i];
  }
  // (the rows' operands are fetched before the barrier: they do not depend on the staged vectors)
  rawo_t bc[IT];
  rawx_t rc[IT];
  {
    const int r = wave;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (lane + 64 * i) * 4;
      const bool ok = c < H && r < S;
      bc[i] = ok ? *reinterpret_cast<const rawo_t *>(blk + (int64_t)r * H + c) : rawo_t{};
      rc[i] = ok ? *reinterpret_cast<const rawx_t *>(res + (int64_t)r * H + c) : rawx_t{};
    }
  }
  __syncthreads();
  for (int r = wave; r < S; r += nwaves) {
    if (r != wave) {
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int c = (lane + 64 * i) * 4;
        bc[i] = c < H ? *reinterpret_cast<const rawo_t *>(blk + (int64_t)r * H + c) : rawo_t{};
        rc[i] = c < H ? *reinterpret_cast<const rawx_t *>(res + (int64_t)r * H + c) : rawx_t{};
      }
    }
    float4 v[IT];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < H) {
        const float4 a = raw_to_f4(bc[i]), rr = raw_to_f4(rc[i]);
        v[i] = make_float4(rr.x + a.x, rr.y + a.y, rr.z + a.z, rr.w + a.w);
        store4<TX>(y + (int64_t)r * H + c, v[i]);
        v[i] = make_float4(to_f32(from_f32<TX>(v[i].x)), to_f32(from_f32<TX>(v[i].y)), to_f32(from_f32<TX>(v[i].z)), to_f32(from_f32<TX>(v[i].w)));
        sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      } else {
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    const float mean = wave_sum(sum) * inv_h(H);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < H) {
        const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
        sq += (a * a + b * b) + (cc * cc + d * d);
      }
    }
    const float rstd = rsqrtf(wave_sum(sq) * inv_h(H) + eps);
    float rsum = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < H) {
        const float4 g4 = sG[lane + 64 * i], b4 = sB[lane + 64 * i];
        const float4 o = make_float4((v[i].x - mean) * rstd * g4.x + b4.x, (v[i].y - mean) * rstd * g4.y + b4.y,
                                     (v[i].z - mean) * rstd * g4.z + b4.z, (v[i].w - mean) * rstd * g4.w + b4.w);
        if (xn_o) store4<TO>(xn_o + (int64_t)r * H + c, o);
        v[i] = make_float4(to_f32(from_f32<TO>(o.x)), to_f32(from_f32<TO>(o.y)), to_f32(from_f32<TO>(o.z)), to_f32(from_f32<TO>(o.w)));
        if constexpr (sizeof(TO) == 2) {
          typedef __attribute__((ext_vector_type(4))) bf16_t bf4;
          const bf4 pk = {(bf16_t)o.x, (bf16_t)o.y, (bf16_t)o.z, (bf16_t)o.w};
          sXN[r * Q + lane + 64 * i] = __builtin_bit_cast(rawo_t, pk);
        } else {
          sXN[r * Q + lane + 64 * i] = __builtin_bit_cast(rawo_t, o);
        }
        rsum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      }
    }
    const float rmean = wave_sum(rsum) * inv_h(H);
    float rsq = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < H) {
        const float a = v[i].x - rmean, b = v[i].y - rmean, cc = v[i].z - rmean, d = v[i].w - rmean;
        rsq += (a * a + b * b) + (cc * cc + d * d);
      }
    }
    const float rrstd = rsqrtf(wave_sum(rsq) * inv_h(H) + reps);
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const bool in = lane + 64 * i < Q;
      const float4 g4 = in ? sRG[lane + 64 * i] : make_float4(0, 0, 0, 0), b4 = in ? sRB[lane + 64 * i] : make_float4(0, 0, 0, 0);
      v[i] = make_float4((v[i].x - rmean) * rrstd * g4.x + b4.x, (v[i].y - rmean) * rrstd * g4.y + b4.y,
                         (v[i].z - rmean) * rrstd * g4.z + b4.z, (v[i].w - rmean) * rrstd * g4.w + b4.w);
    }
    float acc[NN];
#pragma unroll
    for (int n = 0; n < NN; ++n) {
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const float4 wn = lane + 64 * i < Q ? sW[n * Q + lane + 64 * i] : make_float4(0, 0, 0, 0);
        a += (v[i].x * wn.x + v[i].y * wn.y) + (v[i].z * wn.z + v[i].w * wn.w);
      }
      acc[n] = wave_sum(a);
    }
    if (lane < NN) {
      float o = 0.f;
#pragma unroll
      for (int n = 0; n < NN; ++n) if (lane == n) o = acc[n];
      o += rb ? rb[lane] : 0.f;
      logits[r * NN + lane] = o;
      s_lg[r * NN + lane] = o;
    }
  }
  __syncthreads();
  if (t < S) {
    gate_topk_row<NN>(s_lg + t * NN, gates + (int64_t)t * NN, s_idx + t * K, s_w + t * K, NN, K);
    for (int k = 0; k < K; ++k) { idx_o[t * K + k] = s_idx[t * K + k]; w_o[t * K + k] = s_w[t * K + k]; }
  }
  __syncthreads();
  plan_small_body(s_idx, s_w, nullptr, 0, offsets, row_token, row_k, slot_of, S, NN, K, s_off, s_rtok);
  __syncthreads();
  const int rows = s_off[NN];
  for (int r = wave; r < rows; r += nwaves) {
    const int e = expert_of_row(s_off, NN, r);
    const rawo_t *src = sXN + s_rtok[r] * Q;
    float4 v[IT];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      v[i] = lane + 64 * i < Q ? raw_to_f4(src[lane + 64 * i]) : make_float4(0.f, 0.f, 0.f, 0.f);
      sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(sum) * inv_h(H);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int c = (lane + 64 * i) * 4;
      if (c < H) {
        float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
        sq += (a * a + b * b) + (cc * cc + d * d);
      }
    }
    const float rstd = rsqrtf(wave_sum(sq) * inv_h(H) + leps);
    const float4 *ga = sLG + e * Q, *be = sLB + e * Q;
    TO *dst = xg + (int64_t)r * H;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      int c = (lane + 64 * i) * 4;
      if (c < H) {
        float4 g4 = ga[lane + 64 * i], b4 = be[lane + 64 * i];
        float4 o = make_float4((v[i].x - mean) * rstd * g4.x + b4.x, (v[i].y - mean) * rstd * g4.y + b4.y,
                               (v[i].z - mean) * rstd * g4.z + b4.z, (v[i].w - mean) * rstd * g4.w + b4.w);
        store4<TO>(dst + c, o);
      }
    }
    if (lane == 0) { mean_o[r] = mean; rstd_o[r] = rstd; }
  }
}

// ------------------------------------------------------------------------------------------
// Single-token decode step (S <= 16 rows, bf16 activations, fp32 residual stream): the block boundary in front of the SSM block
// - y = res + blk (blk dense, or the MoE combine sum_k wk yr[slot_of] taken on the fly), xn = LayerNorm(y): dropadd_ln_fwd_k's
// arithmetic without dropout - as the PROLOGUE of the in_proj product xz = xn W^T: every work-group of the skinny NT kernel
// (grouped_gemm_nt_skinny_k<KS = 4>, grouped_gemm.hip: 16 output columns per work-group, four waves that split K in
// 32-aligned quarters and meet in LDS in wave order; W rows on the MFMA A operand straight from global memory, requested
// first) normalises the S rows for itself into LDS - no dependency between work-groups, the weight stream stays spread over
// the chip; work-group 0 also writes y.  The same bits as the two launches it replaces.  LN = false: xn [S,H] is given (more than
// a couple of rows: the prologue would be redone by every work-group) and the product reads it from global memory.
// The EPILOGUE, when the step's cache-only half ran ahead (`pre`, decode_step.hip): the xp columns go straight into the conv
// windows and the z columns gate pre - apertis_decode_post's arithmetic on the bf16 values xz would have held; xz is not written.
// ------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) bf16_t dl_bf16x8;
typedef __attribute__((ext_vector_type(4))) float dl_f32x4;

template <int IT, bool LN>
__global__ void __launch_bounds__(256)
decode_ln_inproj_k(const bf16_t *__restrict__ blk, const int32_t *__restrict__ slot_of, const float *__restrict__ wk, int KK,
                   const float *__restrict__ res, const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                   float *__restrict__ y, const bf16_t *__restrict__ xn, const bf16_t *__restrict__ W, int ldw,
                   bf16_t *__restrict__ out, const float *__restrict__ pre, bf16_t *conv_state, int kconv,
                   bf16_t *__restrict__ gated, int Dn, int S, int H, int N) {
  if constexpr (IT <= 4) __builtin_assume(H > 256 * (IT - 1));
  constexpr int KS = 4, U = 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t *sX = reinterpret_cast<bf16_t *>(smem);                 // [S][H]: xn as stored
  __shared__ dl_f32x4 s_part[KS - 1][64];
  const int lane = threadIdx.x & 63, ks = threadIdx.x >> 6, n0 = blockIdx.x * 16;
  const int K = H;
  const int kq = ((K + KS * 32 - 1) / (KS * 32)) * 32;          // this wave's K range: [ks * kq, min(K, ks * kq + kq))
  const int kbeg = ks * kq, kend = min(K, kbeg + kq);
  const int l15 = lane & 15, fg = lane >> 4, kc = fg * 8;
  const int wcol = n0 + l15;
  const bf16_t *wrow = W + (int64_t)min(wcol, N - 1) * ldw;
  const bool w_ok = wcol < N;
  const dl_bf16x8 zero = {};
  dl_bf16x8 a0[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {          // the wave's first batch of W (all of it for K <= 1024)
    const int k = kbeg + u * 32 + kc;
    a0[u] = (k < kend && w_ok) ? *reinterpret_cast<const dl_bf16x8 *>(wrow + k) : zero;
  }
  // ---- the boundary: a wave per row (dropadd_ln_fwd_k, drop_p = 0) ----
  for (int r = ks; LN && r < S; r += KS) {
    float4 v[IT];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < H) {
        float4 a;
        if (slot_of) {
          float4 acc = make_float4(0, 0, 0, 0);
          for (int k = 0; k < KK; ++k) {
            const int slot = slot_of[r * KK + k];
            if (slot < 0) continue;
            const float wv = wk[r * KK + k];
            const float4 t = load4s<bf16_t>(blk + (int64_t)slot * H + c);
            acc.x += t.x * wv; acc.y += t.y * wv; acc.z += t.z * wv; acc.w += t.w * wv;
          }
          a = make_float4(to_f32(from_f32<bf16_t>(acc.x)), to_f32(from_f32<bf16_t>(acc.y)), to_f32(from_f32<bf16_t>(acc.z)), to_f32(from_f32<bf16_t>(acc.w)));
        } else {
          a = load4s<bf16_t>(blk + (int64_t)r * H + c);
        }
        const float4 rr = load4s<float>(res + (int64_t)r * H + c);
        v[i] = make_float4(rr.x + a.x, rr.y + a.y, rr.z + a.z, rr.w + a.w);
        if (blockIdx.x == 0) store4<float>(y + (int64_t)r * H + c, v[i]);
        sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      } else {
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    const float mean = wave_sum(sum) * inv_h(H);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < H) {
        const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
        sq += (a * a + b * b) + (cc * cc + d * d);
      }
    }
    const float rstd = rsqrtf(wave_sum(sq) * inv_h(H) + eps);
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < H) {
        const float4 g4 = load4<float>(gamma + c), b4 = load4<float>(beta + c);
        typedef __attribute__((ext_vector_type(4))) bf16_t bf4;
        const bf4 pk = {(bf16_t)((v[i].x - mean) * rstd * g4.x + b4.x), (bf16_t)((v[i].y - mean) * rstd * g4.y + b4.y),
                        (bf16_t)((v[i].z - mean) * rstd * g4.z + b4.z), (bf16_t)((v[i].w - mean) * rstd * g4.w + b4.w)};
        *reinterpret_cast<bf4 *>(sX + (int64_t)r * H + c) = pk;
      }
    }
  }
  if constexpr (LN) __syncthreads();
  // ---- the product (grouped_gemm_nt_skinny_k<TO, 4>, one block of <= 16 rows) ----
  const bool x_ok = l15 < S;
  const bf16_t *xrow = (LN ? sX : xn) + (int64_t)min(l15, S - 1) * H;
  dl_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = kbeg; k0 < kend; k0 += 32 * U) {
    dl_bf16x8 a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + u * 32 + kc;
      const bool ok = k < kend;
      a[u] = k0 == kbeg ? a0[u] : ((ok && w_ok) ? *reinterpret_cast<const dl_bf16x8 *>(wrow + k) : zero);
      b[u] = (ok && x_ok) ? *reinterpret_cast<const dl_bf16x8 *>(xrow + k) : zero;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[u], b[u], acc, 0, 0, 0);
  }
  if (ks > 0) s_part[ks - 1][lane] = acc;
  __syncthreads();
  if (ks == 0) {
#pragma unroll
    for (int w2 = 0; w2 < KS - 1; ++w2) { const dl_f32x4 t = s_part[w2][lane]; acc[0] += t[0]; acc[1] += t[1]; acc[2] += t[2]; acc[3] += t[3]; }
    const int nq = n0 + fg * 4;
    if (x_ok && nq < N) {
      bf16_t o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = from_f32<bf16_t>(acc[q] + 0.f);
      if (pre) {
        // the step's cache-only half ran ahead (decode_step.hip): xz = (xp | z) is not needed as a tensor - the xp columns are
        // pushed into the conv windows, the z columns gate `pre` (apertis_decode_post's arithmetic on the values as stored)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nq + q;
          if (n < Dn) {
            bf16_t *cs = conv_state + ((int64_t)l15 * Dn + n) * (kconv - 1);
            constexpr int KEEP = 14;
            bf16_t keep[KEEP];
#pragma unroll
            for (int j = 0; j < KEEP; ++j) keep[j] = j + 1 < kconv - 1 ? cs[j + 1] : bf16_t(0);
#pragma unroll
            for (int j = 0; j < KEEP; ++j)
              if (j + 1 < kconv - 1) cs[j] = keep[j];
            cs[kconv - 2] = o[q];
          } else if (n < 2 * Dn) {
            const int c = n - Dn;
            const float zf = to_f32(o[q]);
            gated[(int64_t)l15 * Dn + c] = from_f32<bf16_t>(pre[(int64_t)l15 * Dn + c] * (zf * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-zf * LOG2E_F))));
          }
        }
      } else {
        bf16_t *dst = out + (int64_t)l15 * N + nq;
        if (nq + 3 < N) *reinterpret_cast<uint2 *>(dst) = *reinterpret_cast<const uint2 *>(o);
        else for (int q = 0; q < 4 && nq + q < N; ++q) dst[q] = o[q];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Router backward (dx half) + the boundary's LayerNorm backward in ONE pass over the rows (round 5).  In front of an MoE
// feed-forward the two kernels run back to back on the same rows: router_bwd3_k<MODE 1> writes the total gradient of the
// normalised stream xn ([T,H] in the compute dtype) and layernorm_bwd_k reads it straight back as its `dy` - 2 x 253 MB per
// layer at the bench shape, both kernels at 5-6 TB/s of their own bytes (profiles/r5_probe_row_kernels_nostore.log).  Here a
// wave forms the row of d xn in registers (the router kernel's arithmetic, rounded to the compute dtype exactly where that
// kernel stored it) and carries on with the LayerNorm backward of the same row (layernorm_bwd_k's arithmetic): d y and the
// masked copy d blk are bit-identical to the two-launch form; the four affine-gradient sums are taken in this kernel's row
// order (wave-strided) and folded in a fixed order.  The router's dW / db stay with router_bwd3_k<MODE 2> (96 accumulator
// registers that would put this kernel at one wave per SIMD).
//   y, dres, dx: TX (the residual stream);  xn, grows, dblk: TG (the compute dtype);  no dense gradient term on xn.
//   part_r: [gridDim.x][NN*H + NN + 2H] (this kernel writes the dgamma_r | dbeta_r columns), part_ln: [gridDim.x][2][H].
// ------------------------------------------------------------------------------------------
template <typename TX, typename TG, int IT, int NN>
__global__ void __launch_bounds__(256, 2)
boundary_router_bwd_k(const TX *__restrict__ y, const float *__restrict__ gamma, const float *__restrict__ mean_i,
                      const float *__restrict__ rstd_i, const TX *__restrict__ dres, TX *__restrict__ dx, TG *__restrict__ dblk,
                      float drop_p, uint64_t seed, const TG *__restrict__ xn, const float *__restrict__ rgamma,
                      const float *__restrict__ rbeta, const float *__restrict__ rmean_i, const float *__restrict__ rrstd_i,
                      const float *__restrict__ W, const float *__restrict__ dlogits, const TG *__restrict__ grows,
                      const int32_t *__restrict__ slot_of, int KS, float *__restrict__ part_r, float *__restrict__ part_ln,
                      int64_t T, int H) {
  if constexpr (IT <= 4) __builtin_assume(H > 256 * (IT - 1));
  typedef typename raw4<TX>::type rawx_t;
  typedef typename raw4<TG>::type rawg_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4 *sW = reinterpret_cast<float4 *>(smem);                 // [NN][H/4]
  float4 *red = sW + NN * (H / 4);                               // [2][H/4], one wave at a time
  // the boundary norm's two affine-gradient sums live in LDS, a private [2][H/4] table per wave (with them in registers
  // the kernel does not fit the 256 VGPRs of two waves per SIMD at H = 704, N = 8: 196 bytes of scratch per lane)
  float4 *acc = red + 2 * (H / 4) + (size_t)(threadIdx.x >> 6) * 2 * (H / 4);
  float4 *sG = red + 10 * (H / 4);                               // [2][H/4]: the router norm's gamma, the boundary norm's (as W: read per use)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, Q = H / 4;
  for (int i = threadIdx.x; i < NN * Q; i += 256) sW[i] = reinterpret_cast<const float4 *>(W)[i];
  for (int i = threadIdx.x; i < Q; i += 256) {
    sG[i] = reinterpret_cast<const float4 *>(rgamma)[i];
    sG[Q + i] = reinterpret_cast<const float4 *>(gamma)[i];
  }
  float4 agr[IT], abr[IT];
#pragma unroll
  for (int i = 0; i < IT; ++i) {
    const int c = (lane + 64 * i) * 4;
    (void)c;
    agr[i] = make_float4(0, 0, 0, 0); abr[i] = make_float4(0, 0, 0, 0);
    if (lane + 64 * i < Q) { acc[lane + 64 * i] = make_float4(0, 0, 0, 0); acc[Q + lane + 64 * i] = make_float4(0, 0, 0, 0); }
  }
  __syncthreads();
  const int64_t wave = (int64_t)blockIdx.x * 4 + wv, nw = (int64_t)gridDim.x * 4;
  rawg_t xc[IT], xn_[IT], gc[2][IT], gn_[2][IT];
  rawx_t yc[IT], yn_[IT], rc[IT];
  int sc[2] = {-1, -1}, sn[2] = {-1, -1};
  const bool gath = grows != nullptr;
  auto fetch_slots = [&](int64_t r) -> int { return (gath && r < T && lane < KS) ? slot_of[r * KS + lane] : -1; };
  auto slots_of = [&](int v, int (&so)[2]) {
    so[0] = __builtin_amdgcn_readlane(v, 0);
    so[1] = __builtin_amdgcn_readlane(v, 1);
  };
  int slv_n = -1, slv_n2 = -1;
  auto fetch = [&](rawg_t (&xo)[IT], rawg_t (&go)[2][IT], rawx_t (&yo)[IT], const int (&so)[2], int64_t r) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (lane + 64 * i) * 4;
      const bool ok = c < H && r < T;
      xo[i] = ok ? raw_load(xn + r * H + c) : rawg_t{};
#pragma unroll
      for (int k = 0; k < 2; ++k)
        go[k][i] = (gath && ok && so[k] >= 0) ? raw_load(grows + (int64_t)so[k] * H + c) : rawg_t{};
      yo[i] = ok ? raw_load(y + r * H + c) : rawx_t{};
    }
  };
  // the row's scalars in ONE register (router_bwd3_k): lanes < NN the logit gradients, then the router norm's mean / rstd,
  // then the boundary norm's
  auto fetch_meta = [&](int64_t r) -> float {
    if (r >= T || lane >= NN + 4) return 0.f;
    const float *p = lane < NN ? dlogits + r * NN + lane
                   : lane == NN ? rmean_i + r : lane == NN + 1 ? rrstd_i + r : lane == NN + 2 ? mean_i + r : rstd_i + r;
    return *p;
  };
  auto lane_val = [](float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); };
  float meta = 0.f, meta_next = 0.f;
  if (wave < T) {
    slots_of(fetch_slots(wave), sc);
    slv_n = fetch_slots(wave + nw);
    fetch(xc, gc, yc, sc, wave);
    meta = fetch_meta(wave);
  }
  const uint32_t th = (uint32_t)(drop_p * 65536.f);
  const float ks = 1.f / (1.f - drop_p);
  for (int64_t r = wave; r < T; r += nw) {
    slv_n2 = fetch_slots(r + 2 * nw);
    slots_of(slv_n, sn);
    // (the residual branch's gradient row is met last, ~700 instructions from here: fetched for THIS row, not a row ahead -
    //  a second copy does not fit the 256 VGPRs of two waves per SIMD)
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (lane + 64 * i) * 4;
      rc[i] = (c < H && dres) ? raw_load(dres + r * H + c) : rawx_t{};
    }
    fetch(xn_, gn_, yn_, sn, r + nw);
    meta_next = fetch_meta(r + nw);
    float g[NN];
#pragma unroll
    for (int n = 0; n < NN; ++n) g[n] = lane_val(meta, n);
    const float rmean = lane_val(meta, NN), rrstd = lane_val(meta, NN + 1), mean = lane_val(meta, NN + 2), rstd = lane_val(meta, NN + 3);
    // ---- the router norm + projection, backward (router_bwd3_k<MODE 1>, operation for operation) ----
    float4 xh[IT], dn[IT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const bool in = lane + 64 * i < Q;
      const float4 xv = raw_to_f4(xc[i]);
      const v2f mm = {-rmean, -rmean}, rs = {rrstd, rrstd};
      v2f xh0 = in ? ((v2f){xv.x, xv.y} + mm) * rs : (v2f){0.f, 0.f}, xh1 = in ? ((v2f){xv.z, xv.w} + mm) * rs : (v2f){0.f, 0.f};
      xh[i] = make_float4(xh0.x, xh0.y, xh1.x, xh1.y);
      const float4 gr4 = in ? sG[lane + 64 * i] : make_float4(0, 0, 0, 0);
      const v2f g0 = {gr4.x, gr4.y}, g1 = {gr4.z, gr4.w};
      v2f d0 = {0.f, 0.f}, d1 = {0.f, 0.f};   // dxn = dlogits @ W
#pragma unroll
      for (int n = 0; n < NN; ++n) {
        const float4 wn = in ? sW[n * Q + lane + 64 * i] : make_float4(0, 0, 0, 0);
        const v2f gn = {g[n], g[n]};
        d0 = pk_fma2(gn, (v2f){wn.x, wn.y}, d0); d1 = pk_fma2(gn, (v2f){wn.z, wn.w}, d1);
      }
      v2f a0 = pk_fma2(d0, xh0, (v2f){agr[i].x, agr[i].y}), a1 = pk_fma2(d1, xh1, (v2f){agr[i].z, agr[i].w});
      agr[i] = make_float4(a0.x, a0.y, a1.x, a1.y);
      const v2f b0 = (v2f){abr[i].x, abr[i].y} + d0, b1 = (v2f){abr[i].z, abr[i].w} + d1;
      abr[i] = make_float4(b0.x, b0.y, b1.x, b1.y);
      const v2f dn0 = d0 * g0, dn1 = d1 * g1;
      dn[i] = make_float4(dn0.x, dn0.y, dn1.x, dn1.y);
      s1 += (dn[i].x + dn[i].y) + (dn[i].z + dn[i].w);
      s2 += (dn[i].x * xh[i].x + dn[i].y * xh[i].y) + (dn[i].z * xh[i].z + dn[i].w * xh[i].w);
      __builtin_amdgcn_sched_barrier(0);   // (a chunk at a time: hipcc otherwise fetches every chunk's W rows from LDS up front and spills)
    }
    const float rm1 = wave_sum(s1) * inv_h(H), rm2 = wave_sum(s2) * inv_h(H);
    float4 dq[IT];     // d xn of this row as router_bwd3_k stores it (rounded to the compute dtype): the LayerNorm backward's dy
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (lane + 64 * i) * 4;
      dq[i] = make_float4(0, 0, 0, 0);
      if (c < H) {
        float4 rr = make_float4(0, 0, 0, 0);
        if (gath) {
          float4 ga = make_float4(0, 0, 0, 0);
#pragma unroll
          for (int k = 0; k < 2; ++k)
            if (sc[k] >= 0) { const float4 gv = raw_to_f4(gc[k][i]); ga.x += gv.x; ga.y += gv.y; ga.z += gv.z; ga.w += gv.w; }
          rr.x += to_f32(from_f32<TG>(ga.x)); rr.y += to_f32(from_f32<TG>(ga.y));
          rr.z += to_f32(from_f32<TG>(ga.z)); rr.w += to_f32(from_f32<TG>(ga.w));
        }
        dq[i] = make_float4(to_f32(from_f32<TG>(rrstd * (dn[i].x - rm1 - xh[i].x * rm2) + rr.x)),
                            to_f32(from_f32<TG>(rrstd * (dn[i].y - rm1 - xh[i].y * rm2) + rr.y)),
                            to_f32(from_f32<TG>(rrstd * (dn[i].z - rm1 - xh[i].z * rm2) + rr.z)),
                            to_f32(from_f32<TG>(rrstd * (dn[i].w - rm1 - xh[i].w * rm2) + rr.w)));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- the boundary's LayerNorm, backward (layernorm_bwd_k, operation for operation) ----
    float t1 = 0.f, t2 = 0.f;
    float4 gd[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (lane + 64 * i) * 4;
      const float4 xq = c < H ? raw_to_f4(yc[i]) : make_float4(0, 0, 0, 0), d4 = dq[i];
      xh[i] = make_float4((xq.x - mean) * rstd, (xq.y - mean) * rstd, (xq.z - mean) * rstd, (xq.w - mean) * rstd);
      const float4 g4 = c < H ? sG[Q + lane + 64 * i] : make_float4(0, 0, 0, 0);
      gd[i] = make_float4(d4.x * g4.x, d4.y * g4.y, d4.z * g4.z, d4.w * g4.w);
      if (c < H) {
        float4 ag = acc[lane + 64 * i], ab = acc[Q + lane + 64 * i];
        ag.x += d4.x * xh[i].x; ag.y += d4.y * xh[i].y; ag.z += d4.z * xh[i].z; ag.w += d4.w * xh[i].w;
        ab.x += d4.x; ab.y += d4.y; ab.z += d4.z; ab.w += d4.w;
        acc[lane + 64 * i] = ag; acc[Q + lane + 64 * i] = ab;
        t1 += (gd[i].x + gd[i].y) + (gd[i].z + gd[i].w);
        t2 += (gd[i].x * xh[i].x + gd[i].y * xh[i].y) + (gd[i].z * xh[i].z + gd[i].w * xh[i].w);
      }
    }
    const float m1 = wave_sum(t1) * inv_h(H), m2 = wave_sum(t2) * inv_h(H);
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < H) {
        const float4 rr = raw_to_f4(rc[i]);        // (zeros without dres)
        const float4 dt = make_float4(rstd * (gd[i].x - m1 - xh[i].x * m2) + rr.x, rstd * (gd[i].y - m1 - xh[i].y * m2) + rr.y,
                                      rstd * (gd[i].z - m1 - xh[i].z * m2) + rr.z, rstd * (gd[i].w - m1 - xh[i].w * m2) + rr.w);
        store4<TX>(dx + r * H + c, dt);
        float e[4] = {dt.x, dt.y, dt.z, dt.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = to_f32(from_f32<TX>(e[j]));
        if (drop_p > 0.f) {
          bool keep[4];
          drop_keep4(seed, (uint64_t)r * (uint64_t)H + (uint64_t)c, th, keep);
#pragma unroll
          for (int j = 0; j < 4; ++j) e[j] = keep[j] ? e[j] * ks : 0.f;
        }
        store4<TG>(dblk + r * H + c, make_float4(e[0], e[1], e[2], e[3]));
      }
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) { xc[i] = xn_[i]; gc[0][i] = gn_[0][i]; gc[1][i] = gn_[1][i]; yc[i] = yn_[i]; }
    sc[0] = sn[0]; sc[1] = sn[1];
    slv_n = slv_n2;
    meta = meta_next;
  }
  // block reduction in wave order (the router norm's sums: waves 1..3 take turns in one LDS buffer; the boundary norm's: wave 0
  // adds the four private tables), then one partial row per block and table
  for (int turn = 1; turn < 4; ++turn) {
    __syncthreads();
    if (wv == turn) {
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int cq = lane + 64 * i;
        if (cq < Q) { red[cq] = agr[i]; red[Q + cq] = abr[i]; }
      }
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int i = 0; i < IT; ++i) {
        const int cq = lane + 64 * i;
        if (cq < Q) {
          const float4 u = red[cq], v = red[Q + cq];
          agr[i].x += u.x; agr[i].y += u.y; agr[i].z += u.z; agr[i].w += u.w;
          abr[i].x += v.x; abr[i].y += v.y; abr[i].z += v.z; abr[i].w += v.w;
        }
      }
    }
  }
  if (wv == 0) {
    float *dr = part_r + (int64_t)blockIdx.x * (NN * H + NN + 2 * H) + NN * H + NN;
    float *dl = part_ln + (int64_t)blockIdx.x * 2 * H;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
      const int cq = lane + 64 * i;
      if (cq < Q) {
        float4 ag = acc[cq], ab = acc[Q + cq];
        for (int w = 1; w < 4; ++w) {
          const float4 u = acc[(size_t)w * 2 * Q + cq], v = acc[(size_t)w * 2 * Q + Q + cq];
          ag.x += u.x; ag.y += u.y; ag.z += u.z; ag.w += u.w;
          ab.x += v.x; ab.y += v.y; ab.z += v.z; ab.w += v.w;
        }
        *reinterpret_cast<float4 *>(dr + cq * 4) = agr[i];
        *reinterpret_cast<float4 *>(dr + H + cq * 4) = abr[i];
        *reinterpret_cast<float4 *>(dl + cq * 4) = ag;
        *reinterpret_cast<float4 *>(dl + H + cq * 4) = ab;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Tiny linear: y[T,N] = x[T,:K] W[N,K]^T + b with K <= 64, N <= 16 - the SSM's dt_proj_head
// (Linear(dt_rank -> heads), reference core.py:361,382), whose input is a column slice of the
// x_param_proj output (row stride ldx).  A GEMM library pays ~20 us forward and ~270 us backward
// (a [11 x 98304] x [98304 x 22] weight gradient on 16x16 macro tiles plus a separate bias
// reduction) for 4 MB of traffic.  One row per thread; W and b sit in LDS (broadcast reads).
// Backward: dx per row, and dW/db as per-block partial sums over a row tile staged in LDS
// (entry q < N*K is dW[q], the next N are db; thread p owns q = p, p + 128, ...), folded in a fixed order.
// ------------------------------------------------------------------------------------------
constexpr int TL_MAXK = 64, TL_MAXN = 16, TL_ROWS = 128;

// a thread's K-element row slice -> floats.  VEC: 16-byte loads (row start 16-byte aligned, the slice rounded
// up to whole chunks stays inside the row); lanes hold different rows ~ld apart, so every load instruction
// touches 64 cache lines whatever its width - six 16-byte loads instead of 44 two-byte ones
template <typename TX, bool VEC>
__device__ __forceinline__ void tl_load_row(const TX *row, int K, float (&xr)[TL_MAXK]) {
  constexpr int EPC = 16 / (int)sizeof(TX);
  if constexpr (VEC) {
#pragma unroll
    for (int ch = 0; ch < TL_MAXK / EPC; ++ch) {
      if (ch * EPC < K) {
        float4 lo, hi = make_float4(0, 0, 0, 0);
        if constexpr (sizeof(TX) == 2) {
          const uint4 u = *reinterpret_cast<const uint4 *>(row + ch * EPC);
          lo = raw_to_f4(make_uint2(u.x, u.y));
          hi = raw_to_f4(make_uint2(u.z, u.w));
          const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
          for (int u8 = 0; u8 < 8; ++u8) xr[ch * 8 + u8] = ch * 8 + u8 < K ? v[u8] : 0.f;
        } else {
          lo = *reinterpret_cast<const float4 *>(row + ch * EPC);
          const float v[4] = {lo.x, lo.y, lo.z, lo.w};
#pragma unroll
          for (int u4 = 0; u4 < 4; ++u4) xr[ch * 4 + u4] = ch * 4 + u4 < K ? v[u4] : 0.f;
        }
      } else {
#pragma unroll
        for (int u = 0; u < EPC; ++u) xr[ch * EPC + u] = 0.f;
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < TL_MAXK; ++r) xr[r] = r < K ? to_f32(row[r]) : 0.f;
  }
}

// W (and b) in LDS as a zero-padded [TL_MAXN][TL_MAXK] table read four weights at a time (ds_read_b128, all lanes the
// same address): one LDS instruction per four FMAs instead of one per FMA - the kernels were LDS-issue-bound
__device__ __forceinline__ void tl_stage_w(float *sW, const float *__restrict__ W, int K, int N) {
  for (int i = threadIdx.x; i < TL_MAXN * TL_MAXK; i += TL_ROWS) {
    const int j = i / TL_MAXK, r = i - j * TL_MAXK;
    sW[i] = (j < N && r < K) ? W[j * K + r] : 0.f;
  }
}

template <typename TX, bool VEC>
__global__ void __launch_bounds__(TL_ROWS)
tiny_linear_fwd_k(const TX *__restrict__ x, int64_t ldx, const float *__restrict__ W, const float *__restrict__ b,
                  float *__restrict__ y, int64_t T, int K, int N) {
  __shared__ __attribute__((aligned(16))) float sW[TL_MAXN * TL_MAXK + TL_MAXN];
  tl_stage_w(sW, W, K, N);
  for (int i = threadIdx.x; i < N; i += TL_ROWS) sW[TL_MAXN * TL_MAXK + i] = b ? b[i] : 0.f;
  __syncthreads();
  const float4 *sW4 = reinterpret_cast<const float4 *>(sW);
  for (int64_t t = (int64_t)blockIdx.x * TL_ROWS + threadIdx.x; t < T; t += (int64_t)gridDim.x * TL_ROWS) {
    float xr[TL_MAXK];
    tl_load_row<TX, VEC>(x + t * ldx, K, xr);
    for (int j = 0; j < N; ++j) {
      float a = sW[TL_MAXN * TL_MAXK + j];
#pragma unroll
      for (int r4 = 0; r4 < TL_MAXK / 4; ++r4)
        if (r4 * 4 < K) {    // the pad entries of the last chunk are zeros on both sides
          const float4 w = sW4[j * (TL_MAXK / 4) + r4];
          a = fmaf(xr[4 * r4], w.x, a); a = fmaf(xr[4 * r4 + 1], w.y, a);
          a = fmaf(xr[4 * r4 + 2], w.z, a); a = fmaf(xr[4 * r4 + 3], w.w, a);
        }
      y[t * N + j] = a;
    }
  }
}

// The same for a handful of rows (the decode step: T <= 64): a thread per (row, output) with W straight from global memory -
// the kernel above stages a 16 x 64 table in LDS and then has ONE thread walk all N outputs of a row (13 us for one token).
// Same accumulation chain per output (bias first, then r = 0, 1, ...): the same bits.
template <typename TX>
__global__ void __launch_bounds__(256)
tiny_linear_fwd_small_k(const TX *__restrict__ x, int64_t ldx, const float *__restrict__ W, const float *__restrict__ b,
                        float *__restrict__ y, int64_t T, int K, int N) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= T * N) return;
  const int64_t t = i / N;
  const int j = (int)(i - t * N);
  const TX *row = x + t * ldx;
  const float *w = W + (int64_t)j * K;
  float a = b ? b[j] : 0.f;
  for (int r = 0; r < K; ++r) a = fmaf(to_f32(row[r]), w[r], a);
  y[i] = a;
}

// Backward.  dx per row (W four at a time from LDS, as above).  dW/db as per-block partial sums over the row tile
// staged in LDS: thread p owns the 2 x 4 block dW[2*(p/16) + {0,1}][4*(p%16) + {0..3}] (and db of its two rows when
// p%16 == 0) and reads one 8-byte dy pair and one 16-byte x chunk per row for eight FMAs; rows are walked in order and
// the per-block partials folded in a fixed order.
template <typename TX, bool VEC>
__global__ void __launch_bounds__(TL_ROWS)
tiny_linear_bwd_k(const TX *__restrict__ x, int64_t ldx, const float *__restrict__ W, const float *__restrict__ dy,
                  TX *__restrict__ dx, int64_t lddx, float *__restrict__ part, int64_t T, int K, int N, int zero_to) {
  static_assert(TL_ROWS == (TL_MAXN / 2) * (TL_MAXK / 4), "one 2 x 4 block of dW per thread");
  __shared__ __attribute__((aligned(16))) float sW[TL_MAXN * TL_MAXK];
  __shared__ __attribute__((aligned(16))) float sx[TL_ROWS][TL_MAXK + 4];
  __shared__ __attribute__((aligned(16))) float sdy[TL_ROWS][TL_MAXN + 2];
  tl_stage_w(sW, W, K, N);
  const float4 *sW4 = reinterpret_cast<const float4 *>(sW);
  const int nq = N * K + N;
  const int jb = threadIdx.x >> 4, rb = threadIdx.x & 15;
  float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, accb[2] = {0.f, 0.f};
  for (int64_t t0 = (int64_t)blockIdx.x * TL_ROWS; t0 < T; t0 += (int64_t)gridDim.x * TL_ROWS) {
    const int64_t t = t0 + threadIdx.x;
    const bool live = t < T;
    __syncthreads();   // sW loaded / the previous tile is no longer read
    {
      float xr[TL_MAXK];
      if (live) tl_load_row<TX, VEC>(x + t * ldx, K, xr);
#pragma unroll
      for (int r4 = 0; r4 < TL_MAXK / 4; ++r4)
        *reinterpret_cast<float4 *>(&sx[threadIdx.x][4 * r4]) =
            live ? make_float4(xr[4 * r4], xr[4 * r4 + 1], xr[4 * r4 + 2], xr[4 * r4 + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float dyr[TL_MAXN];
#pragma unroll
    for (int j = 0; j < TL_MAXN; ++j) {
      dyr[j] = (live && j < N) ? dy[t * N + j] : 0.f;
      sdy[threadIdx.x][j] = dyr[j];
    }
    if (live) {
      constexpr int EPC = 16 / (int)sizeof(TX);
      TX *drow = dx + t * lddx;
      for (int r0 = 0; r0 < K; r0 += EPC) {
        float a[EPC];
#pragma unroll
        for (int u = 0; u < EPC; ++u) a[u] = 0.f;
#pragma unroll
        for (int j = 0; j < TL_MAXN; ++j)
          if (j < N) {
#pragma unroll
            for (int u4 = 0; u4 < EPC / 4; ++u4) {
              const float4 w = sW4[j * (TL_MAXK / 4) + (r0 >> 2) + u4];   // zeros past K
              a[4 * u4] = fmaf(dyr[j], w.x, a[4 * u4]); a[4 * u4 + 1] = fmaf(dyr[j], w.y, a[4 * u4 + 1]);
              a[4 * u4 + 2] = fmaf(dyr[j], w.z, a[4 * u4 + 2]); a[4 * u4 + 3] = fmaf(dyr[j], w.w, a[4 * u4 + 3]);
            }
          }
        if (VEC && r0 + EPC <= K) {      // whole 16-byte chunk (the output rows are 16-byte aligned when VEC)
          if constexpr (sizeof(TX) == 2) {
            uint32_t wq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
              wq[u] = (uint32_t)__builtin_bit_cast(uint16_t, from_f32<TX>(a[2 * u])) |
                      ((uint32_t)__builtin_bit_cast(uint16_t, from_f32<TX>(a[2 * u + 1])) << 16);
            *reinterpret_cast<uint4 *>(drow + r0) = make_uint4(wq[0], wq[1], wq[2], wq[3]);
          } else {
            *reinterpret_cast<float4 *>(drow + r0) = make_float4(a[0], a[1], a[2], a[3]);
          }
        } else {
#pragma unroll
          for (int u = 0; u < EPC; ++u)
            if (r0 + u < K) drow[r0 + u] = from_f32<TX>(a[u]);
        }
      }
      // the columns [K, zero_to) behind the row receive zeros: the pad of the projection output's gradient (the caller's
      // split_cols slot next to the dt columns - one strided torch fill per layer otherwise)
      if (zero_to > K) {
        int c = K;
        if (VEC && (K & 3) == 0)
          for (; c + 4 <= zero_to; c += 4) {
            if constexpr (sizeof(TX) == 2) *reinterpret_cast<uint2 *>(drow + c) = make_uint2(0u, 0u);
            else *reinterpret_cast<float4 *>(drow + c) = make_float4(0.f, 0.f, 0.f, 0.f);
          }
        for (; c < zero_to; ++c) drow[c] = from_f32<TX>(0.f);
      }
    }
    __syncthreads();
    if (2 * jb < N && 4 * rb < K) {
      for (int row = 0; row < TL_ROWS; ++row) {
        const float2 d = *reinterpret_cast<const float2 *>(&sdy[row][2 * jb]);
        const float4 xv = *reinterpret_cast<const float4 *>(&sx[row][4 * rb]);
        acc[0][0] = fmaf(d.x, xv.x, acc[0][0]); acc[0][1] = fmaf(d.x, xv.y, acc[0][1]);
        acc[0][2] = fmaf(d.x, xv.z, acc[0][2]); acc[0][3] = fmaf(d.x, xv.w, acc[0][3]);
        acc[1][0] = fmaf(d.y, xv.x, acc[1][0]); acc[1][1] = fmaf(d.y, xv.y, acc[1][1]);
        acc[1][2] = fmaf(d.y, xv.z, acc[1][2]); acc[1][3] = fmaf(d.y, xv.w, acc[1][3]);
        if (rb == 0) { accb[0] += d.x; accb[1] += d.y; }
      }
    }
  }
  float *dst = part + (int64_t)blockIdx.x * nq;
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int j = 2 * jb + jj;
    if (j < N) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (4 * rb + u < K) dst[j * K + 4 * rb + u] = acc[jj][u];
      if (rb == 0) dst[N * K + j] = accb[jj];
    }
  }
}

int check_H(int64_t H) { return (H > 0 && H % 4 == 0 && H <= 256 * 16) ? APERTIS_OK : APERTIS_ERR_UNSUPPORTED; }

}  // namespace

// dispatch a kernel template on IT = ceil(H/256) in {1,2,3,4,6,8,12,16}
#define DISPATCH_IT(H, ...)                                    \
  do {                                                         \
    int it_ = (int)ceil_div64((H), 256);                       \
    if (it_ <= 1) { constexpr int IT = 1; __VA_ARGS__; }              \
    else if (it_ <= 2) { constexpr int IT = 2; __VA_ARGS__; }         \
    else if (it_ <= 3) { constexpr int IT = 3; __VA_ARGS__; }         \
    else if (it_ <= 4) { constexpr int IT = 4; __VA_ARGS__; }         \
    else if (it_ <= 6) { constexpr int IT = 6; __VA_ARGS__; }         \
    else if (it_ <= 8) { constexpr int IT = 8; __VA_ARGS__; }         \
    else if (it_ <= 12) { constexpr int IT = 12; __VA_ARGS__; }       \
    else { constexpr int IT = 16; __VA_ARGS__; }                      \
  } while (0)

#define DISPATCH_2T(da, db, ...)                                                           \
  do {                                                                                     \
    if ((da) == APERTIS_F32 && (db) == APERTIS_F32) { typedef float TA; typedef float TB; __VA_ARGS__; }        \
    else if ((da) == APERTIS_F32 && (db) == APERTIS_BF16) { typedef float TA; typedef bf16_t TB; __VA_ARGS__; } \
    else if ((da) == APERTIS_BF16 && (db) == APERTIS_F32) { typedef bf16_t TA; typedef float TB; __VA_ARGS__; } \
    else if ((da) == APERTIS_BF16 && (db) == APERTIS_BF16) { typedef bf16_t TA; typedef bf16_t TB; __VA_ARGS__; } \
    else return APERTIS_ERR_ARG;                                                           \
  } while (0)

extern "C" int apertis_moe_gate_topk_fwd(const float *logits, float *gates, int32_t *idx, float *w,
                                         int64_t S, int64_t E, int64_t K, void *stream) {
  if (!logits || !gates || !idx || !w || S < 0) return APERTIS_ERR_ARG;
  if (E < 1 || E > MAXE || K < 1 || K > MAXK || K > E) return APERTIS_ERR_UNSUPPORTED;
  if (S == 0) return APERTIS_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)ceil_div64(S, 256)), block(256);
#define GO(EC) hipLaunchKernelGGL(gate_topk_fwd_k<EC>, grid, block, 0, st, logits, gates, idx, w, S, (int)E, (int)K)
  if (E == 4) GO(4); else if (E == 8) GO(8); else if (E == 16) GO(16); else GO(0);
#undef GO
  return apertis_check_launch();
}

extern "C" int apertis_moe_gate_topk_bwd(const float *gates, const int32_t *idx, const float *dw,
                                         const float *dgates, float *dlogits, int64_t S, int64_t E,
                                         int64_t K, void *stream) {
  if (!gates || !idx || !dlogits || S < 0) return APERTIS_ERR_ARG;
  if (E < 1 || E > MAXE || K < 1 || K > MAXK || K > E) return APERTIS_ERR_UNSUPPORTED;
  if (S == 0) return APERTIS_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)ceil_div64(S, 256)), block(256);
#define GO(EC) hipLaunchKernelGGL(gate_topk_bwd_k<EC>, grid, block, 0, st, gates, idx, dw, dgates, dlogits, S, (int)E, (int)K)
  if (E == 4) GO(4); else if (E == 8) GO(8); else if (E == 16) GO(16); else GO(0);
#undef GO
  return apertis_check_launch();
}

extern "C" int64_t apertis_moe_plan_workspace_bytes(int64_t S, int64_t E, int64_t K) {
  int64_t P = E * K, NCH = ceil_div64(S > 0 ? S : 1, 64);
  return (7 * P + 256 * P + 4 + 2 * P * NCH + PLAN_HIST_BLOCKS * P) * 4 + 64;
}

extern "C" int apertis_moe_plan(const int32_t *idx, const float *w, const uint8_t *active,
                                int64_t capacity, int32_t *expert_offsets, int32_t *row_token,
                                int32_t *row_k, int32_t *slot_of, void *ws, int64_t S, int64_t E,
                                int64_t K, void *stream) {
  if (!idx || !w || !expert_offsets || !row_token || !row_k || !slot_of || !ws || S < 0) return APERTIS_ERR_ARG;
  if (E < 1 || E > MAXE || K < 1 || K > MAXK) return APERTIS_ERR_UNSUPPORTED;
  if (S * K > 0x7fffffffLL) return APERTIS_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (S > 0 && S <= 64 && E * K <= 16) {   // a handful of tokens: the whole plan in one launch
    hipLaunchKernelGGL(plan_small_k, dim3(1), dim3(64 * (unsigned)(E * K)), 0, st, idx, w, active, capacity, expert_offsets,
                       row_token, row_k, slot_of, (int)S, (int)E, (int)K);
    return apertis_check_launch();
  }
  PlanWs pw = carve_ws(ws, S > 0 ? S : 1, E, K);
  int nhist = 0;
  if (S > 0) {
    nhist = (int)std::min<int64_t>(ceil_div64(S * K, 1024), PLAN_HIST_BLOCKS);   // four elements per thread where there are enough
    hipLaunchKernelGGL(plan_hist_k, dim3((unsigned)nhist), dim3(256), 0, st, idx, pw.bpart, S * K, (int)E, (int)K);
  }
  hipLaunchKernelGGL(plan_capacity_k, dim3(1), dim3(256), 0, st, pw, active, capacity, expert_offsets, (int)E, (int)K, nhist);
  if (S > 0) {
    if (capacity > 0) {
      if (pw.P <= 32) {
        hipMemsetAsync(pw.ghist, 0, sizeof(int32_t) * (pw.P * 256 + 4), st);   // (the bins and the four arrival counters)
        const unsigned nbh = (unsigned)std::min<int64_t>(ceil_div64(S * K, 1024), 512);
        for (int shift = 24; shift >= 0; shift -= 8)
          hipLaunchKernelGGL(plan_sel_hist_k, dim3(nbh), dim3(256), (size_t)pw.P * 256 * sizeof(int32_t), st, pw, idx, w, S,
                             (int)E, (int)K, shift, 3 - shift / 8);
      } else {
        hipLaunchKernelGGL(plan_select_k, dim3(pw.P), dim3(1024), 0, st, pw, idx, w, S, (int)K);
      }
    }
    dim3 cgrid((unsigned)ceil_div64(pw.NCH, 4)), cblock(256);
    hipLaunchKernelGGL(plan_count_k, cgrid, cblock, 0, st, pw, idx, w, S, (int)E, (int)K);
    hipLaunchKernelGGL(plan_scan_k, dim3(2 * pw.P), dim3(256), 0, st, pw);
    hipLaunchKernelGGL(plan_assign_k, cgrid, cblock, 0, st, pw, idx, w, row_token, row_k, slot_of, S, (int)E, (int)K);
  }
  return apertis_check_launch();
}

extern "C" int apertis_moe_gather_ln_fwd(const void *x, const int32_t *row_token,
                                         const int32_t *expert_offsets, const float *gamma,
                                         const float *beta, float eps, void *xg, float *mean,
                                         float *rstd, int64_t max_rows, int64_t H, int64_t E,
                                         int dtype_x, int dtype_out, void *stream) {
  if (!x || !row_token || !expert_offsets || !gamma || !beta || !xg || !mean || !rstd || max_rows < 0)
    return APERTIS_ERR_ARG;
  if (check_H(H) || E < 1 || E > MAXE) return APERTIS_ERR_UNSUPPORTED;
  if (max_rows == 0) return APERTIS_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)ceil_div64(max_rows, 4)), block(256);
  DISPATCH_2T(dtype_x, dtype_out, DISPATCH_IT(H, hipLaunchKernelGGL((gather_ln_fwd_k<TA, TB, IT>), grid, block, 0, st,
      (const TA *)x, row_token, expert_offsets, gamma, beta, eps, (TB *)xg, mean, rstd, max_rows, (int)H, (int)E)));
  return apertis_check_launch();
}

extern "C" int64_t apertis_moe_gather_ln_bwd_blocks(int64_t max_rows) { return ceil_div64(max_rows > 0 ? max_rows : 1, 4 * GLN_RPW); }

extern "C" int apertis_moe_gather_ln_bwd(const void *x, const int32_t *row_token,
                                         const int32_t *expert_offsets, const float *gamma,
                                         const float *mean, const float *rstd, const void *dxg,
                                         void *dxr, float *dgamma, float *dbeta, float *part, int32_t *blk_expert,
                                         int64_t max_rows, int64_t H, int64_t E, int dtype_x, int dtype_g,
                                         void *stream) {
  // part: workspace [apertis_moe_gather_ln_bwd_blocks(max_rows)][2H] fp32; blk_expert: workspace [same] int32
  if (!x || !row_token || !expert_offsets || !gamma || !mean || !rstd || !dxg || !dgamma || !dbeta || !part || !blk_expert ||
      max_rows < 0)
    return APERTIS_ERR_ARG;   // dxr may be NULL: affine gradients only
  if (check_H(H) || E < 1 || E > MAXE) return APERTIS_ERR_UNSUPPORTED;
  if (max_rows == 0) return APERTIS_OK;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nblk = apertis_moe_gather_ln_bwd_blocks(max_rows);
  dim3 grid((unsigned)nblk), block(256);
  const size_t lds = 3 * 2 * (size_t)H * sizeof(float);
  DISPATCH_2T(dtype_x, dtype_g, DISPATCH_IT(H, hipLaunchKernelGGL((gather_ln_bwd2_k<TA, TB, IT>), grid, block, lds, st,
      (const TA *)x, row_token, expert_offsets, gamma, mean, rstd, (const TB *)dxg, (TB *)dxr, dgamma, dbeta, part,
      blk_expert, max_rows, (int)H, (int)E)));
  hipLaunchKernelGGL(gather_ln_fold_k, dim3((unsigned)ceil_div64(2 * H, 64), (unsigned)E), dim3(1024), 0, st, part, blk_expert,
                     expert_offsets, dgamma, dbeta, max_rows, nblk, (int)H, (int)E);
  return apertis_check_launch();
}

extern "C" int apertis_moe_combine_fwd(const void *yr, const int32_t *slot_of, const float *wk, void *out,
                                       int64_t S, int64_t H, int64_t K, int with_weights, int dtype_yr,
                                       int dtype_out, void *stream) {
  if (!yr || !slot_of || !out || (with_weights && !wk) || S < 0) return APERTIS_ERR_ARG;
  if (check_H(H) || K < 1 || K > MAXK) return APERTIS_ERR_UNSUPPORTED;
  if (S == 0) return APERTIS_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)ceil_div64(S, 4)), block(256);
  DISPATCH_2T(dtype_yr, dtype_out, DISPATCH_IT(H, hipLaunchKernelGGL((combine_fwd_k<TA, TB, IT>), grid, block, 0, st,
      (const TA *)yr, slot_of, wk, (TB *)out, S, (int)H, (int)K, with_weights)));
  return apertis_check_launch();
}

extern "C" int apertis_moe_combine_bwd(const void *dout, const void *yr, const int32_t *row_token,
                                       const int32_t *row_k, const int32_t *expert_offsets, const float *wk,
                                       void *dyr, float *dwk, int64_t max_rows, int64_t S, int64_t H,
                                       int64_t K, int64_t E, int dtype_dout, int dtype_yr, void *stream) {
  if (!dout || !yr || !row_token || !row_k || !expert_offsets || !wk || !dyr || !dwk || max_rows < 0)
    return APERTIS_ERR_ARG;
  if (check_H(H) || K < 1 || K > MAXK || E < 1 || E > MAXE) return APERTIS_ERR_UNSUPPORTED;
  if (max_rows == 0) return APERTIS_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)ceil_div64(max_rows, 4)), block(256);
  DISPATCH_2T(dtype_dout, dtype_yr, DISPATCH_IT(H, hipLaunchKernelGGL((combine_bwd_k<TA, TB, IT>), grid, block, 0, st,
      (const TA *)dout, (const TB *)yr, row_token, row_k, expert_offsets, wk, (TB *)dyr, dwk, max_rows, (int)H,
      (int)K, (int)E)));
  return apertis_check_launch();
}

// ------------------------------------------------------------------------------------------
// Plain LayerNorm over the last dimension (the pre-norms of ApertisAttention / ApertisFeedForward
// and final_post_norm, reference core.py:669,695,847,888,1040,1294) on the same row kernels:
// fp32 residual stream in, compute-dtype (bf16 under autocast) activations out in one pass.
// ------------------------------------------------------------------------------------------
// rows of the LayerNorm backward's workspace: one partial row per block, and behind them the row groups of the two-level fold
constexpr int LN_FOLD_GROUPS = 32;
static int64_t ln_part_rows(int64_t T) { return ceil_div64(T > 0 ? T : 1, 4 * LN_RPW); }
static bool ln_two_level(int64_t nblk) { return nblk >= 8 * LN_FOLD_GROUPS; }
extern "C" int64_t apertis_layernorm_bwd_blocks(int64_t T, int64_t H) {
  (void)H;
  const int64_t nblk = ln_part_rows(T);
  return nblk + (ln_two_level(nblk) ? LN_FOLD_GROUPS : 0);
}

extern "C" int apertis_layernorm_fwd(const void *x, const float *gamma, const float *beta, float eps, void *y,
                                     float *mean, float *rstd, int64_t T, int64_t H, int dtype_x, int dtype_y,
                                     void *stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || T < 0) return APERTIS_ERR_ARG;
  if (check_H(H)) return APERTIS_ERR_UNSUPPORTED;
  if (T == 0) return APERTIS_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)ceil_div64(T, 4)), block(256);
  DISPATCH_2T(dtype_x, dtype_y, DISPATCH_IT(H, hipLaunchKernelGGL((gather_ln_fwd_k<TA, TB, IT>), grid, block, 0, st,
      (const TA *)x, (const int32_t *)nullptr, (const int32_t *)nullptr, gamma, beta, eps, (TB *)y, mean, rstd, T, (int)H,
      1)));
  return apertis_check_launch();
}

extern "C" int apertis_layernorm_bwd(const void *x, const float *gamma, const float *mean, const float *rstd,
                                     const void *dy, const void *dres, void *dx, void *dblk, float drop_p, uint64_t seed,
                                     float *part, float *dgamma, float *dbeta, int64_t T, int64_t H, int dtype_x,
                                     int dtype_g, void *stream) {
  if (!x || !gamma || !mean || !rstd || !dy || !dx || !part || !dgamma || !dbeta || T < 0) return APERTIS_ERR_ARG;
  if (drop_p < 0.f || drop_p >= 1.f) return APERTIS_ERR_ARG;
  if (check_H(H)) return APERTIS_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nblk = ln_part_rows(T);
  dim3 grid((unsigned)nblk), block(256);
  const size_t lds = 3 * 2 * (size_t)H * sizeof(float);
  DISPATCH_2T(dtype_x, dtype_g, DISPATCH_IT(H, hipLaunchKernelGGL((layernorm_bwd_k<TA, TB, IT>), grid, block, lds, st,
      (const TA *)x, gamma, mean, rstd, (const TB *)dy, (const TA *)dres, (TA *)dx, (TB *)dblk, drop_p, seed, part, T, (int)H)));
  const unsigned fx = (unsigned)ceil_div64(2 * H, 64);
  if (ln_two_level(nblk)) {
    float *fold = part + nblk * 2 * H;
    const int64_t rpg = ceil_div64(nblk, LN_FOLD_GROUPS), ng = ceil_div64(nblk, rpg);
    hipLaunchKernelGGL(ln_fold_k, dim3(fx, (unsigned)ng), dim3(1024), 0, st, part, nullptr, nullptr, nblk, (int)H, fold, rpg);
    hipLaunchKernelGGL(ln_fold_k, dim3(fx), dim3(1024), 0, st, fold, dgamma, dbeta, ng, (int)H, nullptr, (int64_t)0);
  } else {
    hipLaunchKernelGGL(ln_fold_k, dim3(fx), dim3(1024), 0, st, part, dgamma, dbeta, nblk, (int)H, nullptr, (int64_t)0);
  }
  return apertis_check_launch();
}

// apertis_layernorm_bwd for a boundary whose block output was the MoE combine (apertis_dropout_add_layernorm_fwd with slot_of),
// with apertis_moe_combine_bwd folded in: dx, dgamma, dbeta as apertis_layernorm_bwd leaves them, dyr [rows, H] and dwk [T, K]
// (pre-zeroed by the caller: dropped slots are not written) as apertis_moe_combine_bwd would from the dblk that is never
// stored.  K <= 2 (APERTIS_ERR_UNSUPPORTED otherwise: call the two entry points).
extern "C" int apertis_layernorm_combine_bwd(const void *x, const float *gamma, const float *mean, const float *rstd,
                                             const void *dy, const void *dres, void *dx, float drop_p, uint64_t seed,
                                             float *part, float *dgamma, float *dbeta, const int32_t *slot_of, const float *wk,
                                             const void *yr, void *dyr, float *dwk, int64_t T, int64_t H, int64_t K,
                                             int dtype_x, int dtype_g, void *stream) {
  if (!x || !gamma || !mean || !rstd || !dy || !dx || !part || !dgamma || !dbeta || !slot_of || !wk || !yr || !dyr || !dwk || T < 0)
    return APERTIS_ERR_ARG;
  if (drop_p < 0.f || drop_p >= 1.f || K < 1) return APERTIS_ERR_ARG;
  if (check_H(H) || K > LN_COMB_K) return APERTIS_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nblk = ln_part_rows(T);
  dim3 grid((unsigned)nblk), block(256);
  const size_t lds = 3 * 2 * (size_t)H * sizeof(float);
  DISPATCH_2T(dtype_x, dtype_g, DISPATCH_IT(H, hipLaunchKernelGGL((layernorm_bwd_k<TA, TB, IT, true>), grid, block, lds, st,
      (const TA *)x, gamma, mean, rstd, (const TB *)dy, (const TA *)dres, (TA *)dx, (TB *)nullptr, drop_p, seed, part, T, (int)H,
      slot_of, wk, (int)K, (const TB *)yr, (TB *)dyr, dwk)));
  const unsigned fx = (unsigned)ceil_div64(2 * H, 64);
  if (ln_two_level(nblk)) {
    float *fold = part + nblk * 2 * H;
    const int64_t rpg = ceil_div64(nblk, LN_FOLD_GROUPS), ng = ceil_div64(nblk, rpg);
    hipLaunchKernelGGL(ln_fold_k, dim3(fx, (unsigned)ng), dim3(1024), 0, st, part, nullptr, nullptr, nblk, (int)H, fold, rpg);
    hipLaunchKernelGGL(ln_fold_k, dim3(fx), dim3(1024), 0, st, fold, dgamma, dbeta, ng, (int)H, nullptr, (int64_t)0);
  } else {
    hipLaunchKernelGGL(ln_fold_k, dim3(fx), dim3(1024), 0, st, part, dgamma, dbeta, nblk, (int)H, nullptr, (int64_t)0);
  }
  return apertis_check_launch();
}

// dispatch on (N, IT): N in {2,4,8,16} compile-time; other N <= 16 are padded by the caller
#define SKINNY_N(N_, ...)                                              \
  do {                                                                 \
    if ((N_) == 2) { constexpr int NN = 2; __VA_ARGS__; }              \
    else if ((N_) == 4) { constexpr int NN = 4; __VA_ARGS__; }         \
    else if ((N_) == 8) { constexpr int NN = 8; __VA_ARGS__; }         \
    else if ((N_) == 16) { constexpr int NN = 16; __VA_ARGS__; }       \
    else return APERTIS_ERR_UNSUPPORTED;                               \
  } while (0)
#define SKINNY_IT(K_, ...)                                             \
  do {                                                                 \
    int it_ = (int)ceil_div64((K_), 256);                              \
    if (it_ <= 1) { constexpr int IT = 1; __VA_ARGS__; }               \
    else if (it_ <= 2) { constexpr int IT = 2; __VA_ARGS__; }          \
    else if (it_ <= 3) { constexpr int IT = 3; __VA_ARGS__; }          \
    else if (it_ <= 4) { constexpr int IT = 4; __VA_ARGS__; }          \
    else return APERTIS_ERR_UNSUPPORTED;                               \
  } while (0)

extern "C" int64_t apertis_skinny_linear_bwd_blocks(int64_t T) { return ceil_div64(T > 0 ? T : 1, 32); }

extern "C" int apertis_skinny_linear_fwd(const void *x, const float *W, const float *b, float *y, int64_t T, int64_t K,
                                         int64_t N, int dtype_x, void *stream) {
  if (!x || !W || !y || T < 0) return APERTIS_ERR_ARG;
  if (K <= 0 || K % 4 || K > 1024 || N < 1 || N > SK_MAXN) return APERTIS_ERR_UNSUPPORTED;
  if (N > 8 && K > 256) return APERTIS_ERR_UNSUPPORTED;   // register budget: N*K/64 weight words per lane
  if (T == 0) return APERTIS_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)std::min<int64_t>(ceil_div64(T, 8), 4096)), block(256);
  if (dtype_x == APERTIS_BF16) {
    SKINNY_N(N, SKINNY_IT(K, hipLaunchKernelGGL((skinny_fwd_k<bf16_t, IT, NN>), grid, block, 0, st, (const bf16_t *)x, W, b, y, T, (int)K)));
  } else if (dtype_x == APERTIS_F32) {
    SKINNY_N(N, SKINNY_IT(K, hipLaunchKernelGGL((skinny_fwd_k<float, IT, NN>), grid, block, 0, st, (const float *)x, W, b, y, T, (int)K)));
  } else return APERTIS_ERR_ARG;
  return apertis_check_launch();
}

extern "C" int apertis_skinny_linear_bwd(const void *x, const float *W, const float *dy, void *dx, float *part,
                                         float *dW_db, int64_t T, int64_t K, int64_t N, int dtype_x, void *stream) {
  // part: workspace [apertis_skinny_linear_bwd_blocks(T)][N*K + N]; dW_db: out [N*K + N] (dW then db)
  if (!x || !W || !dy || !dx || !part || !dW_db || T < 0) return APERTIS_ERR_ARG;
  if (K <= 0 || K % 4 || K > 1024 || N < 1 || N > SK_MAXN) return APERTIS_ERR_UNSUPPORTED;
  if (N > 8 && K > 256) return APERTIS_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nblk = apertis_skinny_linear_bwd_blocks(T);
  dim3 grid((unsigned)nblk), block(256);
  const size_t lds = 3 * (size_t)N * K * sizeof(float);
  if (dtype_x == APERTIS_BF16) {
    SKINNY_N(N, SKINNY_IT(K, hipLaunchKernelGGL((skinny_bwd_k<bf16_t, IT, NN>), grid, block, lds, st, (const bf16_t *)x, W, dy, (bf16_t *)dx, part, T, (int)K)));
  } else if (dtype_x == APERTIS_F32) {
    SKINNY_N(N, SKINNY_IT(K, hipLaunchKernelGGL((skinny_bwd_k<float, IT, NN>), grid, block, lds, st, (const float *)x, W, dy, (float *)dx, part, T, (int)K)));
  } else return APERTIS_ERR_ARG;
  const int64_t cols = N * K + N;
  hipLaunchKernelGGL(fold_rows_k, dim3((unsigned)ceil_div64(cols, 64)), dim3(1024), 0, st, part, dW_db, nblk, cols);
  return apertis_check_launch();
}

extern "C" int64_t apertis_tiny_linear_bwd_blocks(int64_t T) {
  return std::min<int64_t>(ceil_div64(T > 0 ? T : 1, TL_ROWS), 1024);
}

extern "C" int apertis_tiny_linear_fwd(const void *x, int64_t ldx, const float *W, const float *b, float *y, int64_t T,
                                       int64_t K, int64_t N, int dtype_x, void *stream) {
  if (!x || !W || !y || T < 0 || ldx < K) return APERTIS_ERR_ARG;
  if (K < 1 || K > TL_MAXK || N < 1 || N > TL_MAXN) return APERTIS_ERR_UNSUPPORTED;
  if (T == 0) return APERTIS_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)std::min<int64_t>(ceil_div64(T, TL_ROWS), 4096)), block(TL_ROWS);
  if (dtype_x != APERTIS_BF16 && dtype_x != APERTIS_F32) return APERTIS_ERR_ARG;
  if (T <= 64) {   // the decode step
    const dim3 gs((unsigned)ceil_div64(T * N, 256)), bs(256);
    if (dtype_x == APERTIS_BF16) hipLaunchKernelGGL(tiny_linear_fwd_small_k<bf16_t>, gs, bs, 0, st, (const bf16_t *)x, ldx, W, b, y, T, (int)K, (int)N);
    else hipLaunchKernelGGL(tiny_linear_fwd_small_k<float>, gs, bs, 0, st, (const float *)x, ldx, W, b, y, T, (int)K, (int)N);
    return apertis_check_launch();
  }
  const int64_t esz = dtype_x == APERTIS_BF16 ? 2 : 4, epc = 16 / esz;
  const bool vec = (((uintptr_t)x) & 15) == 0 && (ldx * esz) % 16 == 0 && ceil_div64(K, epc) * epc <= ldx;
#define GO(TX, V) hipLaunchKernelGGL((tiny_linear_fwd_k<TX, V>), grid, block, 0, st, (const TX *)x, ldx, W, b, y, T, (int)K, (int)N)
  if (dtype_x == APERTIS_BF16) { if (vec) GO(bf16_t, true); else GO(bf16_t, false); }
  else { if (vec) GO(float, true); else GO(float, false); }
#undef GO
  return apertis_check_launch();
}

extern "C" int apertis_tiny_linear_bwd_pad(const void *x, int64_t ldx, const float *W, const float *dy, void *dx, int64_t lddx,
                                           float *part, float *dW_db, int64_t T, int64_t K, int64_t N, int64_t zero_to, int dtype_x,
                                           void *stream) {
  // part: workspace [apertis_tiny_linear_bwd_blocks(T)][N*K + N]; dW_db: out [N*K + N] (dW then db)
  if (!x || !W || !dy || !dx || !part || !dW_db || T < 0 || ldx < K || lddx < K || zero_to > lddx) return APERTIS_ERR_ARG;
  if (zero_to < K) zero_to = K;
  if (K < 1 || K > TL_MAXK || N < 1 || N > TL_MAXN) return APERTIS_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nblk = apertis_tiny_linear_bwd_blocks(T);
  dim3 grid((unsigned)nblk), block(TL_ROWS);
  if (dtype_x != APERTIS_BF16 && dtype_x != APERTIS_F32) return APERTIS_ERR_ARG;
  const int64_t esz = dtype_x == APERTIS_BF16 ? 2 : 4, epc = 16 / esz;
  const bool vec = (((uintptr_t)x) & 15) == 0 && (ldx * esz) % 16 == 0 && ceil_div64(K, epc) * epc <= ldx &&
                   (((uintptr_t)dx) & 15) == 0 && (lddx * esz) % 16 == 0;
#define GO(TX, V) hipLaunchKernelGGL((tiny_linear_bwd_k<TX, V>), grid, block, 0, st, (const TX *)x, ldx, W, dy, (TX *)dx, lddx, part, T, (int)K, (int)N, (int)zero_to)
  if (dtype_x == APERTIS_BF16) { if (vec) GO(bf16_t, true); else GO(bf16_t, false); }
  else { if (vec) GO(float, true); else GO(float, false); }
#undef GO
  const int64_t cols = N * K + N;
  hipLaunchKernelGGL(fold_rows_k, dim3((unsigned)ceil_div64(cols, 64)), dim3(1024), 0, st, part, dW_db, nblk, cols);
  return apertis_check_launch();
}
extern "C" int apertis_tiny_linear_bwd(const void *x, int64_t ldx, const float *W, const float *dy, void *dx, int64_t lddx,
                                       float *part, float *dW_db, int64_t T, int64_t K, int64_t N, int dtype_x,
                                       void *stream) {
  return apertis_tiny_linear_bwd_pad(x, ldx, W, dy, dx, lddx, part, dW_db, T, K, N, K, dtype_x, stream);
}

extern "C" int64_t apertis_router_bwd_blocks(int64_t T) { return std::min<int64_t>(ceil_div64(T > 0 ? T : 1, 4), 512); }

extern "C" int apertis_router_fwd(const void *x, const float *gamma, const float *beta, float eps, const float *W,
                                  const float *b, float *logits, float *mean, float *rstd, int64_t T, int64_t H, int64_t N,
                                  int dtype_x, void *stream) {
  if (!x || !gamma || !beta || !W || !logits || !mean || !rstd || T < 0) return APERTIS_ERR_ARG;
  if (H <= 0 || H % 4 || H > 1024 || N < 1 || N > 8) return APERTIS_ERR_UNSUPPORTED;
  if (T == 0) return APERTIS_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)std::min<int64_t>(ceil_div64(T, 8), 1024)), block(256);   // every block stages W once
  if (dtype_x == APERTIS_BF16) {
    SKINNY_N(N, SKINNY_IT(H, hipLaunchKernelGGL((router_fwd_k<bf16_t, IT, NN>), grid, block, (size_t)(N * H * 4), st, (const bf16_t *)x, gamma, beta, eps, W, b, logits, mean, rstd, T, (int)H)));
  } else if (dtype_x == APERTIS_F32) {
    SKINNY_N(N, SKINNY_IT(H, hipLaunchKernelGGL((router_fwd_k<float, IT, NN>), grid, block, (size_t)(N * H * 4), st, (const float *)x, gamma, beta, eps, W, b, logits, mean, rstd, T, (int)H)));
  } else return APERTIS_ERR_ARG;
  return apertis_check_launch();
}

extern "C" int apertis_router_bwd_rows(const void *x, const float *gamma, const float *beta, const float *mean,
                                       const float *rstd, const float *W, const float *dlogits, const void *dres,
                                       const void *grows, const int32_t *slot_of, int64_t KS, void *dx, float *part,
                                       float *grads, int64_t T, int64_t H, int64_t N, int dtype_x, void *stream) {
  // part: workspace [apertis_router_bwd_blocks(T)][N*H + N + 2H]; grads: out [N*H dW | N db | H dgamma | H dbeta]
  if (!x || !gamma || !beta || !mean || !rstd || !W || !dlogits || !dx || !part || !grads || T < 0) return APERTIS_ERR_ARG;
  if (grows && (!slot_of || KS < 1)) return APERTIS_ERR_ARG;
  if (grows && KS > 2) return APERTIS_ERR_UNSUPPORTED;   // (the caller then forms the dense gradient with apertis_moe_combine_fwd)
  if (H <= 0 || H % 4 || H > 1024 || N < 1 || N > 8) return APERTIS_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nblk = apertis_router_bwd_blocks(T);
  dim3 grid((unsigned)nblk), block(256);
  const size_t lds3 = (size_t)(2 * N + 2) * H * sizeof(float);
  // wide rows with many outputs: two launches (see router_bwd3_k) - the one-pass kernel is at one wave per SIMD there
  const bool split = N * ((H + 255) / 256) >= 16;
  if (dtype_x == APERTIS_BF16) {
#define ROUTER_BWD(TXT, MODE_) { auto kf = router_bwd3_k<TXT, IT, NN, MODE_>; if (lds3 > 48 * 1024) hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3); hipLaunchKernelGGL(kf, grid, block, lds3, st, (const TXT *)x, gamma, beta, mean, rstd, W, dlogits, (const TXT *)dres, (const TXT *)grows, slot_of, (int)KS, (TXT *)dx, part, T, (int)H); }
    SKINNY_N(N, SKINNY_IT(H, { if (split) { ROUTER_BWD(bf16_t, 1) ROUTER_BWD(bf16_t, 2) } else ROUTER_BWD(bf16_t, 0) }));
  } else if (dtype_x == APERTIS_F32) {
    SKINNY_N(N, SKINNY_IT(H, { if (split) { ROUTER_BWD(float, 1) ROUTER_BWD(float, 2) } else ROUTER_BWD(float, 0) }));
#undef ROUTER_BWD
  } else return APERTIS_ERR_ARG;
  const int64_t cols = N * H + N + 2 * H;
  hipLaunchKernelGGL(fold_rows_k, dim3((unsigned)ceil_div64(cols, 64)), dim3(1024), 0, st, part, grads, nblk, cols);
  return apertis_check_launch();
}

extern "C" int apertis_router_bwd(const void *x, const float *gamma, const float *beta, const float *mean, const float *rstd,
                                  const float *W, const float *dlogits, const void *dres, void *dx, float *part,
                                  float *grads, int64_t T, int64_t H, int64_t N, int dtype_x, void *stream) {
  return apertis_router_bwd_rows(x, gamma, beta, mean, rstd, W, dlogits, dres, nullptr, nullptr, 0, dx, part, grads, T, H, N,
                                 dtype_x, stream);
}

extern "C" int apertis_decode_inproj(const void *blk, const int32_t *slot_of, const float *wk, int64_t KK, const float *res,
                                     const float *gamma, const float *beta, float eps, float *y, const void *xn, const void *W,
                                     int64_t ldw, void *xz, const float *pre, void *conv_state, int64_t kconv, void *gated,
                                     int64_t S, int64_t H, int64_t N, int64_t Dn, void *stream) {
  if (!W) return APERTIS_ERR_ARG;
  if (xn ? false : (!blk || !res || !gamma || !beta || !y || (slot_of && (!wk || KK < 1)))) return APERTIS_ERR_ARG;
  if (pre ? (!conv_state || !gated || N != 2 * Dn || Dn < 1) : !xz) return APERTIS_ERR_ARG;
  if (S < 1 || S > 16 || H < 512 || H % 8 || H > 1024 || N < 4 || N % 4 || ldw < H || ldw % 8) return APERTIS_ERR_UNSUPPORTED;
  if (pre && (kconv < 2 || kconv > 16)) return APERTIS_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)ceil_div64(N, 16)), block(256);
#define DL_GO(LN_) { hipLaunchKernelGGL((decode_ln_inproj_k<IT, LN_>), grid, block, (LN_) ? (size_t)S * H * 2 : (size_t)0, st, (const bf16_t *)blk, \
                                        slot_of, wk, (int)KK, res, gamma, beta, eps, y, (const bf16_t *)xn, (const bf16_t *)W, (int)ldw, \
                                        (bf16_t *)xz, pre, (bf16_t *)conv_state, (int)kconv, (bf16_t *)gated, (int)Dn, (int)S, (int)H, (int)N); }
  if (xn) { SKINNY_IT(H, DL_GO(false)); }
  else { SKINNY_IT(H, DL_GO(true)); }
#undef DL_GO
  return apertis_check_launch();
}

extern "C" int apertis_moe_enter_small(const void *blk, const void *res, const float *gamma, const float *beta, float eps, void *y,
                                       void *xn, const float *rgamma, const float *rbeta, float reps, const float *W,
                                       const float *rb, float *logits, float *gates, int32_t *idx, float *w,
                                       int32_t *expert_offsets, int32_t *row_token, int32_t *row_k, int32_t *slot_of,
                                       const float *lgamma, const float *lbeta, float leps, void *xg, float *mean, float *rstd,
                                       int64_t S, int64_t H, int64_t E, int64_t K, int dtype_x, int dtype_y, void *stream) {
  if (!blk || !res || !gamma || !beta || !y || !rgamma || !rbeta || !W || !logits || !gates || !idx || !w || !expert_offsets ||
      !row_token || !row_k || !slot_of || !lgamma || !lbeta || !xg || !mean || !rstd)
    return APERTIS_ERR_ARG;
  if (S < 1 || S > 16 || (E != 4 && E != 8) || K < 1 || K > E || E * K > 16) return APERTIS_ERR_UNSUPPORTED;
  if (H <= 0 || H % 4 || H > 1024) return APERTIS_ERR_UNSUPPORTED;
  if (dtype_x != APERTIS_F32 || (dtype_y != APERTIS_BF16 && dtype_y != APERTIS_F32)) return APERTIS_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(1), block(64 * (unsigned)(E * K));
  const size_t lds = (size_t)(3 * E + 4) * H * sizeof(float) + (size_t)S * H * (dtype_y == APERTIS_BF16 ? 2 : 4);
  // (+ 4 KiB: the kernel's and plan_small_body's static __shared__ tables share the CU's 160 KiB with the dynamic part;
  //  ops.moe_enter_small_supported mirrors this bound)
  if (lds + 4096 > 160 * 1024) return APERTIS_ERR_UNSUPPORTED;
#define ES_GO(TOT, NN_) { auto kf = moe_enter_small_k<float, TOT, IT, NN_>; \
    if (lds > 48 * 1024) hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(kf, grid, block, lds, st, (const TOT *)blk, (const float *)res, gamma, beta, eps, (float *)y, (TOT *)xn, rgamma, rbeta, \
                       reps, W, rb, logits, gates, idx, w, expert_offsets, row_token, row_k, slot_of, lgamma, lbeta, leps, (TOT *)xg, \
                       mean, rstd, (int)S, (int)K, (int)H); }
  if (dtype_y == APERTIS_BF16) { SKINNY_IT(H, { if (E == 4) ES_GO(bf16_t, 4) else ES_GO(bf16_t, 8) }); }
  else { SKINNY_IT(H, { if (E == 4) ES_GO(float, 4) else ES_GO(float, 8) }); }
#undef ES_GO
  return apertis_check_launch();
}

extern "C" int apertis_moe_route_small(const float *logits, float *gates, int32_t *idx, float *w, int32_t *expert_offsets,
                                       int32_t *row_token, int32_t *row_k, int32_t *slot_of, const void *x, const float *gamma,
                                       const float *beta, float eps, void *xg, float *mean, float *rstd, int64_t S, int64_t H,
                                       int64_t E, int64_t K, int dtype_x, int dtype_xg, void *stream) {
  if (!logits || !gates || !idx || !w || !expert_offsets || !row_token || !row_k || !slot_of || !x || !gamma || !beta || !xg ||
      !mean || !rstd || S < 0)
    return APERTIS_ERR_ARG;
  if (S < 1 || S > 64 || (E != 4 && E != 8 && E != 16) || K < 1 || K > E || K > MAXK || E * K > 16) return APERTIS_ERR_UNSUPPORTED;
  if (check_H(H) || H > 1024) return APERTIS_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(1), block(64 * (unsigned)(E * K));
#define RS_GO(EC_) hipLaunchKernelGGL((moe_route_small_k<TA, TB, IT, EC_>), grid, block, 0, st, logits, gates, idx, w, expert_offsets, \
                                      row_token, row_k, slot_of, (const TA *)x, gamma, beta, eps, (TB *)xg, mean, rstd, (int)S, (int)E, \
                                      (int)K, (int)H)
  DISPATCH_2T(dtype_x, dtype_xg, SKINNY_IT(H, { if (E == 4) RS_GO(4); else if (E == 8) RS_GO(8); else RS_GO(16); }));
#undef RS_GO
  return apertis_check_launch();
}

extern "C" int apertis_boundary_router_bwd(const void *y, const float *gamma, const float *mean, const float *rstd, const void *dres,
                                           void *dx, void *dblk, float drop_p, uint64_t seed, const void *xn, const float *rgamma,
                                           const float *rbeta, const float *rmean, const float *rrstd, const float *W,
                                           const float *dlogits, const void *grows, const int32_t *slot_of, int64_t KS,
                                           float *part, float *rgrads, float *dgamma, float *dbeta, int64_t T, int64_t H,
                                           int64_t N, int dtype_x, int dtype_g, void *stream) {
  // part: workspace [apertis_router_bwd_blocks(T)][N*H + N + 2H  |  2H]  (the router's table, then the boundary norm's);
  // rgrads: out [N*H dW | N db | H dgamma_r | H dbeta_r]; dgamma / dbeta [H]: the boundary norm's
  if (!y || !gamma || !mean || !rstd || !dx || !dblk || !xn || !rgamma || !rbeta || !rmean || !rrstd || !W || !dlogits || !part ||
      !rgrads || !dgamma || !dbeta || T < 0)
    return APERTIS_ERR_ARG;
  if (drop_p < 0.f || drop_p >= 1.f) return APERTIS_ERR_ARG;
  if (grows && (!slot_of || KS < 1)) return APERTIS_ERR_ARG;
  if (grows && KS > 2) return APERTIS_ERR_UNSUPPORTED;
  if (H <= 0 || H % 4 || H > 1024 || N < 1 || N > 8) return APERTIS_ERR_UNSUPPORTED;
  if (dtype_x != APERTIS_F32 || (dtype_g != APERTIS_BF16 && dtype_g != APERTIS_F32)) return APERTIS_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nblk = apertis_router_bwd_blocks(T), cols = N * H + N + 2 * H;
  float *part_ln = part + nblk * cols;
  dim3 grid((unsigned)nblk), block(256);
  const size_t ldsf = (size_t)(N + 2 + 8 + 2) * H * sizeof(float), lds3 = (size_t)(2 * N + 2) * H * sizeof(float);
#define BR_BWD(TGT) { auto kf = boundary_router_bwd_k<float, TGT, IT, NN>; auto k2 = router_bwd3_k<TGT, IT, NN, 2>; \
    if (lds3 > 48 * 1024) hipFuncSetAttribute((const void *)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3); \
    if (ldsf > 48 * 1024) hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf); \
    hipLaunchKernelGGL(kf, grid, block, ldsf, st, (const float *)y, gamma, mean, rstd, (const float *)dres, (float *)dx, (TGT *)dblk, drop_p, seed, \
                       (const TGT *)xn, rgamma, rbeta, rmean, rrstd, W, dlogits, (const TGT *)grows, slot_of, (int)KS, part, part_ln, T, (int)H); \
    hipLaunchKernelGGL(k2, grid, block, lds3, st, (const TGT *)xn, rgamma, rbeta, rmean, rrstd, W, dlogits, (const TGT *)nullptr, \
                       (const TGT *)nullptr, (const int32_t *)nullptr, 0, (TGT *)nullptr, part, T, (int)H); }
  if (dtype_g == APERTIS_BF16) { SKINNY_N(N, SKINNY_IT(H, BR_BWD(bf16_t))); }
  else { SKINNY_N(N, SKINNY_IT(H, BR_BWD(float))); }
#undef BR_BWD
  hipLaunchKernelGGL(fold_rows_k, dim3((unsigned)ceil_div64(cols, 64)), dim3(1024), 0, st, part, rgrads, nblk, cols);
  hipLaunchKernelGGL(ln_fold_k, dim3((unsigned)ceil_div64(2 * H, 64)), dim3(1024), 0, st, part_ln, dgamma, dbeta, nblk, (int)H,
                     (float *)nullptr, (int64_t)0);
  return apertis_check_launch();
}

extern "C" int64_t apertis_moe_gate_aux_blocks(int64_t S) { return ceil_div64(S > 0 ? S : 1, 256); }

extern "C" int apertis_moe_gate_topk_noisy_aux_fwd(const float *logits, const float *w_noise, float alpha, uint64_t seed,
                                                   float *gates, int32_t *idx, float *w, float *lse, float *part, float *stats,
                                                   int64_t S, int64_t E, int64_t K, float lb_coef, float rz_coef, void *stream) {
  // part: workspace [apertis_moe_gate_aux_blocks(S)][2E+1]; stats: out [2 + E] = [lb, rz, frac_e]; w_noise NULL: no noise
  if (!logits || !gates || !idx || !w || !lse || !part || !stats || S <= 0) return APERTIS_ERR_ARG;
  if (E < 1 || E > MAXE || K < 1 || K > MAXK || K > E) return APERTIS_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nblk = apertis_moe_gate_aux_blocks(S);
  dim3 grid((unsigned)nblk), block(256);
#define GO(EC) hipLaunchKernelGGL(gate_topk_aux_fwd_k<EC>, grid, block, 0, st, logits, w_noise, alpha, seed, gates, idx, w, lse, part, S, (int)E, (int)K)
  if (E == 4) GO(4); else if (E == 8) GO(8); else if (E == 16) GO(16); else GO(0);
#undef GO
  hipLaunchKernelGGL(gate_aux_fold_k, dim3(1), dim3(1024), 0, st, part, stats, nblk, S, (int)E, lb_coef, rz_coef);
  return apertis_check_launch();
}

extern "C" int apertis_moe_gate_topk_aux_fwd(const float *logits, float *gates, int32_t *idx, float *w, float *lse,
                                             float *part, float *stats, int64_t S, int64_t E, int64_t K, float lb_coef,
                                             float rz_coef, void *stream) {
  return apertis_moe_gate_topk_noisy_aux_fwd(logits, nullptr, 0.f, 0, gates, idx, w, lse, part, stats, S, E, K, lb_coef, rz_coef,
                                             stream);
}

extern "C" int apertis_moe_gate_topk_noisy_aux_bwd(const float *gates, const int32_t *idx, const float *dw, const float *lse,
                                                   const float *stats, const float *dlb, const float *drz, float lb_coef,
                                                   float rz_coef, const float *w_noise, float alpha, uint64_t seed,
                                                   float *dlogits, float *npart, float *dw_noise, int64_t S, int64_t E,
                                                   int64_t K, void *stream) {
  // w_noise != NULL: npart = workspace [apertis_moe_gate_aux_blocks(S)][E], dw_noise = out [E]
  if (!gates || !idx || !lse || !stats || !dlogits || S < 0 || (w_noise && (!npart || !dw_noise))) return APERTIS_ERR_ARG;
  if (!w_noise) npart = nullptr;
  if (E < 1 || E > MAXE || K < 1 || K > MAXK || K > E) return APERTIS_ERR_UNSUPPORTED;
  if (S == 0) return APERTIS_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)ceil_div64(S, 256)), block(256);
#define GO(EC) hipLaunchKernelGGL(gate_topk_aux_bwd_k<EC>, grid, block, 0, st, gates, idx, dw, lse, stats, dlb, drz, lb_coef, rz_coef, dlogits, npart, seed, S, (int)E, (int)K)
  if (E == 4) GO(4); else if (E == 8) GO(8); else if (E == 16) GO(16); else GO(0);
#undef GO
  if (npart) hipLaunchKernelGGL(gate_noise_fold_k, dim3(1), dim3(1024), 0, st, npart, w_noise, alpha, dw_noise, (int64_t)grid.x, (int)E);
  return apertis_check_launch();
}

extern "C" int apertis_moe_gate_topk_aux_bwd(const float *gates, const int32_t *idx, const float *dw, const float *lse,
                                             const float *stats, const float *dlb, const float *drz, float lb_coef,
                                             float rz_coef, float *dlogits, int64_t S, int64_t E, int64_t K, void *stream) {
  return apertis_moe_gate_topk_noisy_aux_bwd(gates, idx, dw, lse, stats, dlb, drz, lb_coef, rz_coef, nullptr, 0.f, 0, dlogits,
                                             nullptr, nullptr, S, E, K, stream);
}

extern "C" int apertis_dropout_add_layernorm_fwd(const void *blk, const int32_t *slot_of, const float *wk, int64_t K,
                                                 const void *res, const float *gamma, const float *beta, float eps, void *y,
                                                 void *xn, float *mean, float *rstd, int64_t T, int64_t H, float drop_p,
                                                 uint64_t seed, int dtype_x, int dtype_y, void *stream) {
  if (!blk || !res || !gamma || !beta || !y || !xn || !mean || !rstd || T < 0 || drop_p < 0.f || drop_p >= 1.f)
    return APERTIS_ERR_ARG;
  if (slot_of && (!wk || K < 1 || K > MAXK)) return APERTIS_ERR_ARG;
  if (check_H(H)) return APERTIS_ERR_UNSUPPORTED;
  if (T == 0) return APERTIS_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)ceil_div64(T, 4)), block(256);
  DISPATCH_2T(dtype_x, dtype_y, DISPATCH_IT(H, hipLaunchKernelGGL((dropadd_ln_fwd_k<TA, TB, IT>), grid, block, 0, st,
      (const TB *)blk, slot_of, wk, (int)K, (const TA *)res, gamma, beta, eps, (TA *)y, (TB *)xn, mean, rstd, T, (int)H, drop_p,
      seed)));
  return apertis_check_launch();
}

extern "C" int apertis_dropout_add_layernorm_router_fwd(const void *blk, const void *res, const float *gamma, const float *beta,
                                                        float eps, void *y, void *xn, float *mean, float *rstd,
                                                        const float *rgamma, const float *rbeta, float reps, const float *W,
                                                        const float *rb, float *logits, float *rmean, float *rrstd, int64_t T,
                                                        int64_t H, int64_t N, float drop_p, uint64_t seed, int dtype_x,
                                                        int dtype_y, void *stream) {
  if (!blk || !res || !gamma || !beta || !y || !xn || !mean || !rstd || !rgamma || !rbeta || !W || !logits || !rmean || !rrstd ||
      T < 0 || drop_p < 0.f || drop_p >= 1.f)
    return APERTIS_ERR_ARG;
  if (H <= 0 || H % 4 || H > 1024 || (N != 2 && N != 4 && N != 8)) return APERTIS_ERR_UNSUPPORTED;
  if (T == 0) return APERTIS_OK;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)(N + 4) * H * sizeof(float);
  dim3 grid((unsigned)std::min<int64_t>(ceil_div64(T, 4), 1024)), block(256);   // persistent waves: four per SIMD resident
#define BR_GO(NN_) { auto kf = dropadd_ln_router_fwd_k<TA, TB, IT, NN_>; if (lds > 48 * 1024) hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(kf, grid, block, lds, st, (const TB *)blk, (const TA *)res, gamma, beta, eps, (TA *)y, (TB *)xn, mean, rstd, rgamma, rbeta, reps, W, rb, logits, rmean, rrstd, T, (int)H, drop_p, seed); }
  DISPATCH_2T(dtype_x, dtype_y, SKINNY_IT(H, { if (N == 2) BR_GO(2) else if (N == 4) BR_GO(4) else BR_GO(8) }));
#undef BR_GO
  return apertis_check_launch();
}
