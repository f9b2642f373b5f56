// Grouped GEMM on CDNA4 matrix cores (gfx950) for the MoE expert MLPs, and with one group the
// patch-embed / vision-projection GEMMs.
//
// Reference: the per-expert nn.Linear calls of AdaptiveExpertSystem
// (/root/reference/src/model/core.py:434-442,596) - 16 small addmm launches per layer there,
// one launch per GEMM here.  Rows are expert-sorted (apertis_moe_plan); group e owns rows
// [offsets[e], offsets[e+1]).  Tiles never straddle a group; the grid is sized from an upper
// bound on rows and every work-group finds its (group, m-tile) from the device-side offsets,
// so no host sync is needed.
//
// Tile: 128 (rows) x 128 (cols) x 128 B of K per step (64 bf16 / 32 fp32), 256 threads =
// 2x2 waves, each wave 64x64 = 4x4 MFMA tiles of 16x16.
//   bf16: v_mfma_f32_16x16x32_bf16  (8 k per lane per instruction, fp32 accumulate)
//   fp32: v_mfma_f32_16x16x4_f32    (exact fp32 FMA chain; the parity path)
// The WEIGHT tile feeds the MFMA "A" operand and the ACTIVATION tile the "B" operand, so the
// accumulator holds 4 consecutive output columns per lane (D row = n, D col = m): the epilogue
// writes 8/16-byte pieces of output rows into an LDS staging tile and the tile leaves as whole
// 16-byte-per-lane row segments.
// LDS image of both operand tiles: [row][8 x 16-byte chunk], chunk index XOR (row & 7), which
// makes the ds_read_b128 fragment reads conflict-free (rows are 128 B, two per bank row).
// Global -> LDS goes through registers (one tile in flight, written to the other buffer after
// the MFMAs of the current one: one barrier per K step).
#include "common.h"
#include <algorithm>
#include <type_traits>

namespace {

constexpr int BM = 128, BN = 128;
constexpr int ROWB = 128;            // bytes of K per tile row
constexpr int TILE_BYTES = BM * ROWB;  // 16 KiB per operand tile
constexpr int NT = 256;

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) bf16_t bf16x8;

// ---- Dropout mask of the expert MLP's activation (reference core.py:439: Dropout(p) between the two Linears; every kernel of
// this file, forward and backward, draws it from here).  16 random bits per element; elements (row, 2c) and (row, 2c + 1) of an
// [*, N] tensor take the low / high half of
//     h(row, c) = fin(rowmix(row) ^ colmix(c)),   fin(x) = x * 0x2C1B3C6D, x ^= x >> 15
// - FACTORISED (round 5): rowmix and colmix are full murmur3 finalisers of (seed, row) resp. (seed, column pair), computed once
// per row resp. once per lane and tile by the epilogues that walk whole tiles; what is left per element pair is one xor, one
// multiply and one xor-shift.  (Rounds 1-4: one full finaliser - two multiplies, three xor-shifts - of the linear pair index per
// element pair: 63 of the 182 VALU instructions of the saved-gradient epilogue's row piece.)  The multiply between the xor of
// the two halves and the output breaks the xor-linearity h(r1,c1)^h(r1,c2)^h(r2,c1)^h(r2,c2) = 0 of the bare xor;
// tests/test_moe_kernels_gpu.py checks keep fraction and row / column correlations of the recovered mask.
__device__ __forceinline__ uint32_t gd_fmix(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ uint32_t gd_rowmix(uint64_t seed, uint64_t row) {
  return gd_fmix((uint32_t)row * 0x9E3779B9u + (uint32_t)(row >> 32) * 0x7F4A7C15u + (uint32_t)seed);
}
__device__ __forceinline__ uint32_t gd_colmix(uint64_t seed, uint32_t cpair) {
  return gd_fmix(cpair * 0x85EBCA77u ^ (uint32_t)(seed >> 32));
}
__device__ __forceinline__ uint32_t gd_pair(uint32_t rowmix, uint32_t colmix) {
  uint32_t x = rowmix ^ colmix;
  x *= 0x2C1B3C6Du;
  x ^= x >> 15;
  return x;
}
__device__ __forceinline__ bool gd_keep(uint64_t seed, int64_t row, int64_t col, uint32_t thresh16) {
  const uint32_t h = gd_pair(gd_rowmix(seed, (uint64_t)row), gd_colmix(seed, (uint32_t)(col >> 1)));
  return ((col & 1) ? (h >> 16) : (h & 0xffffu)) >= thresh16;
}
// four consecutive elements from a column that is a multiple of 4
__device__ __forceinline__ void gd_keep4(uint64_t seed, int64_t row, int64_t col0, uint32_t thresh16, bool (&keep)[4]) {
  const uint32_t rm = gd_rowmix(seed, (uint64_t)row), c = (uint32_t)(col0 >> 1);
  const uint32_t h0 = gd_pair(rm, gd_colmix(seed, c)), h1 = gd_pair(rm, gd_colmix(seed, c + 1u));
  keep[0] = (h0 & 0xffffu) >= thresh16; keep[1] = (h0 >> 16) >= thresh16;
  keep[2] = (h1 & 0xffffu) >= thresh16; keep[3] = (h1 >> 16) >= thresh16;
}

template <typename T> struct frag_t;
template <> struct frag_t<bf16_t> { typedef bf16x8 type; };
template <> struct frag_t<float> { typedef f32x4 type; };

// 16-byte output store of the epilogues: non-temporal.  The outputs of a layer's GEMMs are 0.3 - 1.2 GB each; written through
// the caches they only displace what the next kernel reads (+1 % on the step by themselves, A/B inside one gpurun call; the
// row kernels' stores are non-temporal for the same reason, moe_routing.hip)
__device__ __forceinline__ void out_store16(void *p, uint4 v) {
  typedef __attribute__((ext_vector_type(4))) unsigned u4;
  u4 o = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(o, reinterpret_cast<u4 *>(p));
}
__device__ __forceinline__ void mma(f32x4 &acc, const bf16x8 &a, const bf16x8 &b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
// fp32: a 16-byte chunk holds 4 k-values of this lane's row; MFMA step j consumes element j
// (both operands use the same k assignment, so the dot product is complete and exact fp32)
__device__ __forceinline__ void mma(f32x4 &acc, const f32x4 &a, const f32x4 &b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc, 0, 0, 0);
}

// erf: exact (ocml erff) on the fp32 parity path; on the bf16 path Abramowitz-Stegun 7.1.26
// (max abs error 1.5e-7, far below bf16's 2^-8) with hardware exp2/rcp - the ocml erff costs more
// VALU time than the whole K loop of a K=704 tile.
template <bool FAST> __device__ __forceinline__ float erf_t(float x) {
  if constexpr (!FAST) {
    return erff(x);
  } else {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-ax * ax * LOG2E_F);
    const float r = 1.f - p * t * e;
    return copysignf(r, x);
  }
}
template <bool FAST> __device__ __forceinline__ float exp_t(float x) {
  if constexpr (FAST) return __builtin_amdgcn_exp2f(x * LOG2E_F);
  else return expf(x);
}

// GELU on the bf16 path (round 4).  gelu(x) = x * Phi(x), gelu'(x) = Phi(x) + x * phi(x), with
//   Phi(-|x|) = erfc(|x| / sqrt2) / 2 = (a1 t + a2 t^2 + a3 t^3) / 2 * exp(-x^2 / 2),  t = 1 / (1 + p |x| / sqrt2)
// (Abramowitz-Stegun 7.1.25, three terms, |erf error| <= 2.5e-5; rounds 1-3 used the five-term 7.1.26).  The result is
// rounded to bf16 (2^-9) and is evaluated on a pre-activation that was itself rounded to bf16, which moves Phi in the tail by
// x^2 * 2^-9 relative - more than the approximation does anywhere.  exp(-x^2/2) serves phi(x) as well: ONE rcp and ONE exp2
// per element yield both functions.  The caller's output scale s (1 / (1 - p) of the dropout) is folded into the
// coefficients, so scaled outputs cost no multiply: q = s * Phi(-|x|), P = s * Phi(x) = x >= 0 ? s - q : q,
//   s * gelu(x) = x * P,   s * gelu'(x) = P + (x * s / sqrt(2 pi)) * exp(-x^2/2).
// Every form below performs the same IEEE operations in the same order (scalar, or two elements per v_pk_* instruction):
// the fused and the stand-alone epilogues agree bit for bit.  (13 -> 10 lane operations for both functions against round
// 3's, besides the two transcendentals; the epilogue of the saved-gradient forward is VALU-bound.)
struct GeluK { float c1, c2, c3, s, sphi; };
__device__ __forceinline__ GeluK gelu_consts(float s) {
  return {s * (0.5f * 0.3480242f), s * (0.5f * -0.0958798f), s * (0.5f * 0.7478556f), s, s * 0.3989422804014327f};
}
#define GELU_TK (0.47047f * 0.70710678118654752f)
// P = s * Phi(x), e = exp(-x^2 / 2)
__device__ __forceinline__ void gelu_terms(float x, const GeluK &k, float &P, float &e) {
  const float t = __builtin_amdgcn_rcpf(fmaf(GELU_TK, fabsf(x), 1.f));
  float p = fmaf(k.c3, t, k.c2);
  p = fmaf(p, t, k.c1);
  e = __builtin_amdgcn_exp2f((x * x) * (-0.5f * LOG2E_F));
  const float q = (p * t) * e;
  P = x >= 0.f ? k.s - q : q;
}
__device__ __forceinline__ float gelu_fast(float x, const GeluK &k) {
  float P, e;
  gelu_terms(x, k, P, e);
  return x * P;
}
__device__ __forceinline__ float gelu_grad_fast(float x, const GeluK &k) {
  float P, e;
  gelu_terms(x, k, P, e);
  return fmaf(x * k.sphi, e, P);
}
// the same, two elements per instruction where the packed fp32 pipe has one (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: the same
// IEEE operation per element, in the same order - bit-identical to the scalar forms); rcp, exp2 and the select stay per element
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat2(float v) { return (v2f){v, v}; }
__device__ __forceinline__ void gelu_both_fast2(v2f x, const GeluK &k, v2f &g, v2f &dg) {
  const v2f ax = {fabsf(x.x), fabsf(x.y)};
  const v2f ta = pk_fma(splat2(GELU_TK), ax, splat2(1.f));
  const v2f t = {__builtin_amdgcn_rcpf(ta.x), __builtin_amdgcn_rcpf(ta.y)};
  v2f p = pk_fma(splat2(k.c3), t, splat2(k.c2));
  p = pk_fma(p, t, splat2(k.c1));
  const v2f ea = (x * x) * splat2(-0.5f * LOG2E_F);
  const v2f e = {__builtin_amdgcn_exp2f(ea.x), __builtin_amdgcn_exp2f(ea.y)};
  const v2f q = (p * t) * e;
  const v2f smq = splat2(k.s) - q;
  const v2f P = {x.x >= 0.f ? smq.x : q.x, x.y >= 0.f ? smq.y : q.y};
  g = x * P;
  dg = pk_fma(x * splat2(k.sphi), e, P);
}

// a row piece (four pairs) stage by stage: the four dependent chains side by side in program order - hipcc schedules a single
// chain at a time otherwise and pads the dependent v_pk_* pairs with s_nop (60 per four rows of the saved-gradient epilogue)
__device__ __forceinline__ void gelu_both_fast8(const v2f (&x)[4], const GeluK &k, v2f (&g)[4], v2f (&dg)[4]) {
  v2f t[4], p[4], e[4], q[4];
  // (1 + k |x| per element with the |.| as a source modifier: from the packed form hipcc makes two v_and and a v_pk_fma_f32)
  const float tk = GELU_TK;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    asm("v_fma_f32 %0, |%1|, %2, 1.0" : "=v"(t[w].x) : "v"(x[w].x), "s"(tk));
    asm("v_fma_f32 %0, |%1|, %2, 1.0" : "=v"(t[w].y) : "v"(x[w].y), "s"(tk));
  }
#pragma unroll
  for (int w = 0; w < 4; ++w) e[w] = (x[w] * x[w]) * splat2(-0.5f * LOG2E_F);
#pragma unroll
  for (int w = 0; w < 4; ++w) t[w] = (v2f){__builtin_amdgcn_rcpf(t[w].x), __builtin_amdgcn_rcpf(t[w].y)};
#pragma unroll
  for (int w = 0; w < 4; ++w) e[w] = (v2f){__builtin_amdgcn_exp2f(e[w].x), __builtin_amdgcn_exp2f(e[w].y)};
#pragma unroll
  for (int w = 0; w < 4; ++w) p[w] = pk_fma(splat2(k.c3), t[w], splat2(k.c2));
#pragma unroll
  for (int w = 0; w < 4; ++w) p[w] = pk_fma(p[w], t[w], splat2(k.c1));
#pragma unroll
  for (int w = 0; w < 4; ++w) q[w] = p[w] * t[w];
#pragma unroll
  for (int w = 0; w < 4; ++w) q[w] = q[w] * e[w];
#pragma unroll
  for (int w = 0; w < 4; ++w) p[w] = splat2(k.s) - q[w];
#pragma unroll
  for (int w = 0; w < 4; ++w) p[w] = (v2f){x[w].x >= 0.f ? p[w].x : q[w].x, x[w].y >= 0.f ? p[w].y : q[w].y};
#pragma unroll
  for (int w = 0; w < 4; ++w) g[w] = x[w] * p[w];
#pragma unroll
  for (int w = 0; w < 4; ++w) t[w] = x[w] * splat2(k.sphi);
#pragma unroll
  for (int w = 0; w < 4; ++w) dg[w] = pk_fma(t[w], e[w], p[w]);
}

template <bool FAST> __device__ __forceinline__ float act_fwd(float x, int act) {
  switch (act) {
    case APERTIS_ACT_GELU:
      if constexpr (FAST) return gelu_fast(x, gelu_consts(1.f));
      else return 0.5f * x * (1.f + erf_t<FAST>(x * 0.70710678118654752f));
    case APERTIS_ACT_RELU: return x > 0.f ? x : 0.f;
    case APERTIS_ACT_SILU: return x / (1.f + exp_t<FAST>(-x));
    default: return x;
  }
}
template <bool FAST> __device__ __forceinline__ float act_grad(float x, int act) {
  switch (act) {
    case APERTIS_ACT_GELU:
      if constexpr (FAST) return gelu_grad_fast(x, gelu_consts(1.f));
      else return 0.5f * (1.f + erf_t<FAST>(x * 0.70710678118654752f)) + x * 0.3989422804014327f * exp_t<FAST>(-0.5f * x * x);
    case APERTIS_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case APERTIS_ACT_SILU: { float s = 1.f / (1.f + exp_t<FAST>(-x)); return s * (1.f + x * (1.f - s)); }
    default: return 1.f;
  }
}

struct TileCoord { int e, m0, rows_left; int64_t row0; bool valid; };

// m-tile index -> (group, first row in group, rows left); mt counts tiles over all groups.
// The E+1 offsets are fetched by E+1 lanes in one go and scanned from LDS (a serial chain of
// dependent global loads here costs about a microsecond per tile).
__device__ __forceinline__ TileCoord find_tile(const int32_t *offsets, int E, int mt, int32_t *s_off, int tid) {
  for (int i = tid; i <= E; i += NT) s_off[i] = offsets[i];
  __syncthreads();
  TileCoord t; t.valid = false;
  int acc = 0;
  for (int e = 0; e < E; ++e) {
    int r0 = s_off[e], r1 = s_off[e + 1];
    int nt = (r1 - r0 + BM - 1) / BM;
    if (mt < acc + nt) {
      t.e = e; t.m0 = (mt - acc) * BM; t.row0 = (int64_t)r0 + t.m0; t.rows_left = r1 - r0 - t.m0; t.valid = true;
      return t;
    }
    acc += nt;
  }
  return t;
}

// XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a
// contiguous run of tiles so neighbouring tiles (same activation rows, next weight columns)
// meet in one L2.  Bijective for any grid size.
// two floats -> one dword of bf16 (one v_cvt_pk_bf16_f32; a conversion per element wastes half of each)
__device__ __forceinline__ uint32_t pack_bf16x2_any(float a, float b) {
  typedef float f2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
  const f2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2_t));
}
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, loc = bid / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// Order of the (m-tile, n-tile) pairs of a grid of MT x NTL tiles: m-tiles in groups of G; inside a group the n-tiles in
// panels of NB; inside a panel m outer, n inner.  Tiles that share an activation row block are then neighbours (they start
// together and read it in step: one L2 fill for NB tiles) and tiles that share a weight tile are NB apart, i.e. the XCD's
// ~64 resident work-groups of the two-per-CU kernel work under ONE weight panel of NB x 128 rows that stays in its L2, where
// the n-fastest walk over all NTL n-tiles streamed the whole W[e] (3.96 MB at N=2816, K=704 = the XCD's entire L2) past
// every 256-row block.  G bounds how far apart the NTL/NB passes over an activation block are (it comes back from the
// Infinity Cache).  G <= 0 or NB >= NTL: the n-fastest walk.
__device__ __forceinline__ void tile_walk(int t, int MT, int NTL, int G, int NB, int &mt, int &nt) {
  if (G <= 0 || NB <= 0 || NB >= NTL) { mt = t / NTL; nt = t - mt * NTL; return; }
  const int g = t / (G * NTL), r = t - g * G * NTL;
  const int gg = min(G, MT - g * G);           // m-tiles of this group (the last one may be short)
  const int per = gg * NB, pn = r / per, rr = r - pn * per;
  const int wp = min(NB, NTL - pn * NB);       // width of this panel (the last one may be narrow)
  const int mi = rr / wp;
  mt = g * G + mi;
  nt = pn * NB + (rr - mi * wp);
}

template <typename T>
__device__ __forceinline__ void load_tile_regs(uint4 (&regs)[4], const T *base, int64_t ld, int rows_valid,
                                               int k0, int K, int tid) {
  constexpr int KPC = 16 / sizeof(T);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int q = tid + i * NT;
    int row = q >> 3, c = q & 7;
    int k = k0 + c * KPC;
    if (row < rows_valid && k < K)
      regs[i] = *reinterpret_cast<const uint4 *>(base + (int64_t)row * ld + k);
    else
      regs[i] = make_uint4(0, 0, 0, 0);
  }
}

__device__ __forceinline__ void store_tile_lds(char *lds, const uint4 (&regs)[4], int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int q = tid + i * NT;
    int row = q >> 3, c = q & 7;
    *reinterpret_cast<uint4 *>(lds + row * ROWB + ((c ^ (row & 7)) << 4)) = regs[i];
  }
}

// dgrad epilogue fusion: one 16-byte output chunk of dh is turned into dpre = dh * keep/(1-p) *
// act'(pre) with the matching chunk of the saved pre-activation (what apertis_act_dropout_bwd does
// as a separate pass over three [rows, N] tensors)
template <typename TO, bool FAST, int ACT = -1, bool DROP = true>
__device__ __forceinline__ uint4 actbwd_chunk(uint4 dhc, uint4 prec, int64_t row, int64_t col0, int64_t N, int act_rt,
                                             float drop_p_rt, uint64_t seed, float keep_scale, uint32_t thresh16) {
  // ACT >= 0 fixes the activation (and DROP the dropout) at compile time: a per-value runtime switch breaks the
  // instruction stream into branchy blocks that cost more than the arithmetic
  const int act = ACT >= 0 ? ACT : act_rt;
  const float drop_p = ACT >= 0 ? (DROP ? 1.f : 0.f) : drop_p_rt;
  if constexpr (sizeof(TO) == 2) {   // bf16: unpacked with shifts, no byte-addressed temporaries
    const uint32_t a[4] = {dhc.x, dhc.y, dhc.z, dhc.w}, b[4] = {prec.x, prec.y, prec.z, prec.w};
    uint32_t o[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bool keep[4] = {true, true, true, true};
      if (drop_p > 0.f) gd_keep4(seed, row, col0 + 4 * h, thresh16, keep);
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const uint32_t aw = a[2 * h + w], bw = b[2 * h + w];
        const float g0 = __builtin_bit_cast(float, aw << 16) * act_grad<FAST>(__builtin_bit_cast(float, bw << 16), act);
        const float g1 = __builtin_bit_cast(float, aw & 0xffff0000u) * act_grad<FAST>(__builtin_bit_cast(float, bw & 0xffff0000u), act);
        const TO r0 = from_f32<TO>(keep[2 * w] ? g0 * keep_scale : 0.f), r1 = from_f32<TO>(keep[2 * w + 1] ? g1 * keep_scale : 0.f);
        o[2 * h + w] = (uint32_t)__builtin_bit_cast(uint16_t, r0) | ((uint32_t)__builtin_bit_cast(uint16_t, r1) << 16);
      }
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
  } else {
    constexpr int NE = 16 / sizeof(TO);
    union { uint4 u; TO e[NE]; } a, b, o;
    a.u = dhc; b.u = prec;
    bool keep[4] = {true, true, true, true};
    if (drop_p > 0.f) gd_keep4(seed, row, col0, thresh16, keep);
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const float g = to_f32(a.e[j]) * act_grad<FAST>(to_f32(b.e[j]), act);
      o.e[j] = from_f32<TO>(keep[j] ? g * keep_scale : 0.f);
    }
    return o.u;
  }
}

template <typename T, typename TO>
__global__ void __launch_bounds__(NT)
grouped_gemm_nt_k(const T *__restrict__ X, const T *__restrict__ W, const float *__restrict__ bias,
                  const int32_t *__restrict__ offsets, TO *__restrict__ C, TO *__restrict__ pre_act,
                  const TO *__restrict__ mul_pre, int N, int K, int ldw, int E, int n_tiles, int act, float drop_p,
                  uint64_t seed) {
  typedef typename frag_t<T>::type frag;
  constexpr int KPC = 16 / sizeof(T);
  constexpr int BK = ROWB / sizeof(T);
  constexpr int CPITCH = BN * sizeof(TO) + 16;  // padded C staging pitch
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = tile / n_tiles, ntile = tile - mt * n_tiles;
  const TileCoord tc = find_tile(offsets, E, mt, reinterpret_cast<int32_t *>(smem), tid);
  if (!tc.valid) return;
  __syncthreads();  // offsets scratch is about to be overwritten by the first tiles
  const int n0 = ntile * BN;
  const int rows_valid = min(BM, tc.rows_left);
  const int cols_valid = min(BN, N - n0);
  const T *xbase = X + tc.row0 * K;
  const T *wbase = W + ((int64_t)tc.e * N + n0) * ldw;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (K + BK - 1) / BK;
  const int frow = lane & 15, fg = lane >> 4;
  auto compute_tile = [&](const char *xs, const char *ws) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      frag wf[4], xf[4];
      const int chunk = kk * 4 + fg;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int wrow = wn * 64 + i * 16 + frow;
        wf[i] = *reinterpret_cast<const frag *>(ws + wrow * ROWB + ((chunk ^ (wrow & 7)) << 4));
        int xrow = wm * 64 + i * 16 + frow;
        xf[i] = *reinterpret_cast<const frag *>(xs + xrow * ROWB + ((chunk ^ (xrow & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mma(acc[i][j], wf[i], xf[j]);
    }
  };
  {
    uint4 xr[4], wr[4];
    load_tile_regs<T>(xr, xbase, K, rows_valid, 0, K, tid);
    load_tile_regs<T>(wr, wbase, ldw, cols_valid, 0, K, tid);
    store_tile_lds(smem, xr, tid);
    store_tile_lds(smem + TILE_BYTES, wr, tid);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      char *xs = smem + (kt & 1) * 2 * TILE_BYTES;
      char *ws = xs + TILE_BYTES;
      if (kt + 1 < nk) {
        load_tile_regs<T>(xr, xbase, K, rows_valid, (kt + 1) * BK, K, tid);
        load_tile_regs<T>(wr, wbase, ldw, cols_valid, (kt + 1) * BK, K, tid);
      }
      compute_tile(xs, ws);
      if (kt + 1 < nk) {
        char *xn = smem + ((kt + 1) & 1) * 2 * TILE_BYTES;
        store_tile_lds(xn, xr, tid);
        store_tile_lds(xn + TILE_BYTES, wr, tid);
      }
      __syncthreads();
    }
  }

  // epilogue: acc[i][j] = D tile (n-subtile i, m-subtile j); lane: m = frow, n = fg*4 + reg
  const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const uint32_t thresh16 = (uint32_t)(drop_p * 65536.f);
  float bv[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int n = n0 + wn * 64 + i * 16 + fg * 4 + r;
      bv[i][r] = (bias && n < N) ? bias[(int64_t)tc.e * N + n] : 0.f;
    }
  constexpr int CPR = BN * sizeof(TO) / 16;  // 16-byte chunks per output row
  auto flush_tile = [&](TO *dst, const TO *mulp) {
    __syncthreads();
    for (int q = tid; q < BM * CPR; q += NT) {
      int row = q / CPR, c = q % CPR;
      int ncol = c * (16 / (int)sizeof(TO));
      if (row < rows_valid && ncol < cols_valid) {
        const int64_t g = (tc.row0 + row) * N + n0 + ncol;
        uint4 v = *reinterpret_cast<const uint4 *>(smem + row * CPITCH + c * 16);
        if (mulp)
          v = actbwd_chunk<TO, sizeof(T) == 2>(v, *reinterpret_cast<const uint4 *>(mulp + g), tc.row0 + row, n0 + ncol,
                                                N, act, drop_p, seed, keep_scale, thresh16);
        out_store16(dst + g, v);
      }
    }
    __syncthreads();
  };
  if (pre_act) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int m = wm * 64 + j * 16 + frow, n = wn * 64 + i * 16 + fg * 4;
        TO *p = reinterpret_cast<TO *>(smem + m * CPITCH) + n;
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = from_f32<TO>(acc[i][j][r] + bv[i][r]);
      }
    flush_tile(pre_act, nullptr);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int m = wm * 64 + j * 16 + frow, n = wn * 64 + i * 16 + fg * 4;
      TO *p = reinterpret_cast<TO *>(smem + m * CPITCH) + n;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[i][j][r] + bv[i][r];
        if (!mul_pre) {
          v = to_f32(from_f32<TO>(v));  // the activation sees the pre-activation as stored (bf16-rounded under bf16)
          v = act_fwd<sizeof(T) == 2>(v, act);
          if (drop_p > 0.f)
            v = gd_keep(seed, tc.row0 + m, n0 + n + r, thresh16) ? v * keep_scale : 0.f;
        }
        p[r] = from_f32<TO>(v);
      }
    }
  flush_tile(C, mul_pre);
}

// ------------------------------------------------------------------------------------------
// Big-tile NT kernel: 256 x 256 x 64 (bf16), 512 threads = 2(M) x 4(N) waves, each wave 128 x 64
// = 8 x 4 MFMA tiles of 16x16.  A 128^2 tile needs ~64 flop per L2 byte, i.e. ~39 TB/s from L2 at
// the MFMA peak - more than the chip's L2 can deliver; 256^2 halves that.  Operands arrive by
// LDS-DMA (global_load_lds, 16 B/lane) into a double-buffered 2 x 64 KiB ring, one barrier per K
// step (the next tile's DMA is in flight during the MFMAs of the current one).  Same swizzle and
// operand roles as the 128^2 kernel.  One work-group per CU (128 KiB LDS); the kernel itself is the persistent
// grouped_gemm_nt256p_k below (a one-tile-per-work-group form came first and was removed).
// ------------------------------------------------------------------------------------------
constexpr int BM2 = 256, BN2 = 256, NT2 = 512;
constexpr int TILE2_BYTES = BM2 * ROWB;  // 32 KiB per operand tile

// LDS-DMA through inline asm.  hipcc tracks an LDS-DMA builtin as a store to LDS and puts
// `s_waitcnt vmcnt(0)` in front of the next LDS read it cannot prove disjoint - every
// ds_read_b64_tr_b16 - which serialises the double buffer (the DMA of step k+1 would be waited for
// before the MFMAs of step k start).  The ordering is done by hand here (vmcnt(0) + barrier at the
// top of each step), so the DMA is kept out of the compiler's sight.
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4i raw_buffer_rsrc(const void *base, uint32_t bytes) {
  const uint64_t a = (uint64_t)base;
  v4i r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu));   // stride 0: raw buffer
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);                              // num_records: reads past it return 0
  r[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ uint32_t lds_addr_of(const void *p) {
  return (uint32_t)(size_t)(__attribute__((address_space(3))) const char *)p;
}
// 64 lanes x 16 B from buffer offset voff (per lane) to LDS lds_addr + 16 * lane (lds_addr wave-uniform)
__device__ __forceinline__ void lds_dma16(const v4i &rs, uint32_t lds_addr, uint32_t voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(__builtin_amdgcn_readfirstlane((int)lds_addr)), "v"(voff), "s"(rs) : "memory", "m0");
}
// same, from a per-lane global address
__device__ __forceinline__ void lds_dma16_global(const void *src, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
               :: "s"(__builtin_amdgcn_readfirstlane((int)lds_addr)), "v"(src) : "memory", "m0");
}

// ------------------------------------------------------------------------------------------
// Persistent 256 x 256 NT kernel.  With one tile per work-group a K=704 tile took 39 us, ~24 in the K
// loop: every CU ends its K loop at the same time, so a round's epilogue is one chip-wide burst of
// output stores with the matrix pipes idle, and the K loops run with HBM idle.  Here one work-group
// per CU walks its tiles and the output stores drain under the next tile's MFMAs.  vmcnt retires in
// order, so a wave that has stores in flight cannot wait for a younger LDS-DMA without also waiting
// for the stores; the roles are therefore split by wave: waves 4-7 ("store waves") issue all output
// stores of a tile and then sit out the DMA for the first `solo` K steps of the next tile (waves
// 0-3 issue 16 pieces each instead of 8), never waiting on vmcnt meanwhile; at step `solo` they
// drain and rejoin.  The next tile's step-0 DMA is issued (by waves 0-3) BEFORE the epilogue, so
// the per-tile prologue latency disappears as well.  The epilogue converts the accumulators and
// stages them through ring buffer 1 in two 128-row halves.
// ------------------------------------------------------------------------------------------
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// Work-group barrier for LDS traffic only.  __syncthreads() is a release/acquire fence as well: hipcc puts
// `s_waitcnt vmcnt(0)` in front of it, i.e. every output store of the wave has to reach memory before the
// barrier - exactly the latency the kernels below hide.  Here only this wave's LDS operations are drained
// (LDS-DMA arrivals are waited for explicitly with wait_vmcnt where a barrier publishes them).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct PTile { int valid, e, rows_valid, n0, cols_valid; int64_t row0; };

// Dynamic tile queue of the persistent kernel (queue != NULL): after its first (static, XCD-affine) tile a work-group
// takes tile indices from device counters, so a work-group that starts late - its CU was held by another kernel, e.g. an
// RCCL collective on the communication stream - simply finds the queues drained instead of walking a full static share on
// its own after everybody else has finished.  ONE COUNTER PER XCD (NTQ_STRIDE ints apart): a work-group first draws from
// its own XCD's counter, whose tickets enumerate exactly the tiles the static walk gives that XCD (round r >= 1, position
// j * 8 + xcd), so neighbouring tiles still meet in one L2 - with a single counter for the whole chip they did not, and
// the queue cost 3-10 % when nothing competed for the CUs; only when its own counter has run past the last tile does a
// work-group steal from the next XCD's.  The counters are owned by the caller (stream-owned workspace, NTQ_INTS ints); the
// entry point zeroes them on the launch stream in front of the kernel, so nothing survives a launch - no process-global
// state, and an aborted launch cannot leave a dirty slot behind.
constexpr int NTQ_STRIDE = 64, NTQ_INTS = 8 * NTQ_STRIDE;
// next tile index (>= G) for a work-group of XCD `home`, or n_valid when every queue is drained
__device__ __forceinline__ int ntq_next(int *queue, int home, int G, int n_valid) {
  const int per = (G + 7) / 8;
  for (int s = 0; s < 8; ++s) {
    const int x = (home + s) & 7;
    while (true) {
      const int k = atomicAdd(queue + x * NTQ_STRIDE, 1);
      const int r = 1 + k / per, within = (k - (r - 1) * per) * 8 + x;
      const int64_t t = (int64_t)r * G + within;
      if (within >= G) { if ((int64_t)r * G >= n_valid) break; continue; }   // a position past a ragged G: skip it
      if (t < n_valid) return (int)t;
      break;   // this XCD's tiles are gone (tickets grow with k)
    }
  }
  return n_valid;
}
// raw = the pre-activation pass, r = round; ACT >= 0 fixes the activation and DROP the dropout at
// compile time (a per-value runtime switch costs more than the conversion itself), ACT < 0 = runtime
template <typename TO, bool raw, int r, int ACT, bool DROP>
__device__ __forceinline__ void nt256p_out_round(const f32x4 (&acc)[4][8], const float (&bv)[4][4], TO *__restrict__ dst,
                                                const TO *__restrict__ mul_pre, char *stg, const PTile &cur, int N, int act,
                                                float drop_p, uint64_t seed, float keep_scale, uint32_t thresh16, int tid,
                                                int wm, int wn, int frow, int fg) {
  static_assert(sizeof(TO) == 2, "staging layout assumes 2-byte outputs");
  // two rounds through the 64 KiB of ring buffer 1: round r carries the m-subtiles j in [4r, 4r+4) of
  // BOTH wave rows, i.e. 128 tile rows x 512 B, 16-byte chunks XOR-swizzled with the row (no padding
  // left in the buffer); every wave converts in every round, the store waves issue 16 stores per thread
  {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int j = r * 4 + jj;
        const int srow = wm * 64 + jj * 16 + frow, m = wm * 128 + j * 16 + frow;
        const int chunk = wn * 8 + i * 2 + (fg >> 1);
        uint32_t o[4];
        bool keep[4] = {true, true, true, true};
        if (!raw && (ACT >= 0 ? DROP : (!mul_pre && drop_p > 0.f)))
          gd_keep4(seed, cur.row0 + m, cur.n0 + wn * 64 + i * 16 + fg * 4, thresh16, keep);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float a = acc[i][j][q];
          asm volatile("" : "+v"(a));   // pins the conversion inside its round
          float v = a + bv[i][q];
          if (!raw && (ACT >= 0 || !mul_pre)) {
            if (ACT != APERTIS_ACT_NONE) v = act_fwd<true>(to_f32(from_f32<TO>(v)), ACT >= 0 ? ACT : act);
            v = keep[q] ? v * keep_scale : 0.f;   // keep_scale is 1 without dropout
          }
          o[q] = __builtin_bit_cast(uint16_t, from_f32<TO>(v));
        }
        *reinterpret_cast<uint2 *>(stg + srow * 512 + ((chunk ^ frow) << 4) + (fg & 1) * 8) =
            make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
      }
    __syncthreads();
    if (tid >= NT2 / 2) {   // store waves
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int c4 = it * (NT2 / 2) + tid - NT2 / 2;
      const int srow = c4 >> 5, c = c4 & 31;
      const int row = (srow >> 6) * 128 + (r * 4 + ((srow >> 4) & 3)) * 16 + (srow & 15);
      const int ncol = c * 8;
      if (row < cur.rows_valid && ncol < cur.cols_valid) {
        const int64_t g = (cur.row0 + row) * N + cur.n0 + ncol;
        uint4 v = *reinterpret_cast<const uint4 *>(stg + srow * 512 + ((c ^ (srow & 15)) << 4));
        if (ACT < 0 && mul_pre && !raw)
          v = actbwd_chunk<TO, true>(v, *reinterpret_cast<const uint4 *>(mul_pre + g), cur.row0 + row, cur.n0 + ncol, N, act,
                                     drop_p, seed, keep_scale, thresh16);
        out_store16(dst + g, v);
      }
    }
    }
    __syncthreads();
  }
}

// RAGGED: K % 64 != 0 or W has its own row pitch (the dense projections); otherwise ldw == K and both
// operands share one lane offset
template <typename TO, bool RAGGED>
__global__ void __launch_bounds__(NT2)
grouped_gemm_nt256p_k(const bf16_t *__restrict__ X, const bf16_t *__restrict__ W, const float *__restrict__ bias,
                      const int32_t *__restrict__ offsets, TO *__restrict__ C, TO *__restrict__ pre_act,
                      const TO *__restrict__ mul_pre, int N, int K, int ldw, int E, int n_tiles, int total_tiles,
                      int solo, int act, float drop_p, uint64_t seed, int *__restrict__ queue) {
  typedef bf16_t T;
  typedef bf16x8 frag;
  constexpr int BK = 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *stg = smem + 2 * TILE2_BYTES;                // ring buffer 1 doubles as the C staging area
  int32_t *s_off = reinterpret_cast<int32_t *>(smem + 4 * TILE2_BYTES);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int frow = lane & 15, fg = lane >> 4;
  const int G = gridDim.x;
  for (int i = tid; i <= E; i += NT2) s_off[i] = offsets[i];
  __syncthreads();

  // the valid tiles are the first n_valid of the launch grid's (m-tile, n-tile) enumeration
  int mt_valid = 0;
  for (int e = 0; e < E; ++e) mt_valid += (s_off[e + 1] - s_off[e] + BM2 - 1) / BM2;
  const int n_valid = __builtin_amdgcn_readfirstlane(min(total_tiles, mt_valid * n_tiles));

  const int nk = (K + BK - 1) / BK;
  const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const uint32_t thresh16 = (uint32_t)(drop_p * 65536.f);

  f32x4 acc[4][8];
  PTile cur; cur.valid = 0; cur.e = 0; cur.rows_valid = 0; cur.n0 = 0; cur.cols_valid = 0; cur.row0 = 0;
  // one pass of this loop = [find + prefetch tile i] [epilogue of tile i-1] [K loop of tile i]; every
  // piece of code has a single call site
  int *s_next = s_off + 1026;   // (E <= 1024: the offsets end at s_off[1024])
  for (int t = blockIdx.x;;) {
    PTile nxt; nxt.valid = 0;
    if (t < n_valid) {
      const int r = t / G, within = t - r * G, gr = min(G, n_valid - r * G);
      const int tile = r * G + xcd_remap(within, gr);
      const int mt = tile / n_tiles, ntile = tile - mt * n_tiles;
      int accm = 0;
      for (int e = 0; e < E; ++e) {
        const int r0 = s_off[e], r1 = s_off[e + 1];
        const int nt = (r1 - r0 + BM2 - 1) / BM2;
        if (mt < accm + nt) {
          const int m0 = (mt - accm) * BM2;
          // wave-uniform by construction: pin the fields to SGPRs (LDS loads land in VGPRs)
          nxt.valid = 1;
          nxt.e = __builtin_amdgcn_readfirstlane(e);
          nxt.row0 = (int64_t)__builtin_amdgcn_readfirstlane(r0 + m0);
          nxt.rows_valid = __builtin_amdgcn_readfirstlane(min(BM2, r1 - r0 - m0));
          nxt.n0 = __builtin_amdgcn_readfirstlane(ntile * BN2);
          nxt.cols_valid = __builtin_amdgcn_readfirstlane(min(BN2, N - ntile * BN2));
          break;
        }
        accm += nt;
      }
    }
    // the next tile's first DMA goes out before the previous tile's outputs are touched.  Operands
    // come through raw buffer descriptors sized to the tile's valid rows: the hardware bounds check
    // zero-fills ragged rows, and a piece's address is one per-lane VGPR (row-in-piece and swizzled
    // 16-byte chunk, identical for all pieces) + a scalar piece/K-step offset
    // K need not be a multiple of 64: the last K step then reads past a row's end - the next row's
    // first columns, or zeros past the tile's last valid row (the K-step offset is part of the
    // range-checked lane offset) - and W carries zero columns there (row pitch ldw >= the K steps)
    const int ldb = K * (int)sizeof(T), ldwb = ldw * (int)sizeof(T);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T *>(X + nxt.row0 * K), 0, nxt.rows_valid * ldb, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T *>(W + ((int64_t)nxt.e * N + nxt.n0) * ldw), 0, nxt.cols_valid * ldwb, 0x00020000);
    // a piece's address = one per-lane VGPR per operand (row-in-piece and swizzled 16-byte chunk) + scalar
    // terms.  The hardware range check covers the lane offset only, so the piece's row offset always
    // goes there; the K-step offset rides in the (unchecked) scalar offset unless K is ragged
    const int chunk_sw = ((lane & 7) ^ (lane >> 3)) << 4;
    const int voffx0 = (lane >> 3) * ldb + chunk_sw, voffw0 = RAGGED ? (lane >> 3) * ldwb + chunk_sw : voffx0;
    auto piece = [&](char *xs, int p, int kt) {   // 8 rows x 128 B of both operands
      const int kbytes = kt * BK * (int)sizeof(T);
      // (a runtime select in the ragged variant: hipcc spills 75 VGPRs when it is folded, 25 when it is not)
      const bool ktail = RAGGED && (K & (BK - 1)) != 0;
      const int kv = ktail ? kbytes : 0, ks = ktail ? 0 : kbytes;
      const int vx = voffx0 + (p * 8 * ldb + kv), vw = RAGGED ? voffw0 + (p * 8 * ldwb + kv) : vx;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void *)(xs + p * 1024), 16, vx, ks, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void *)(xs + TILE2_BYTES + p * 1024), 16,
                                               vw, ks, 0, 0);
    };
    // K step kt's operands (32 + 32 pieces): by waves 0-3 alone while kt <= solo
    auto stage = [&](int buf, int kt) {
      char *xs = smem + buf * 2 * TILE2_BYTES;
      if (kt > solo) {
#pragma unroll
        for (int j = 0; j < 4; ++j) piece(xs, wave * 4 + j, kt);
      } else if (wave < 4) {
#pragma unroll
        for (int j = 0; j < 8; ++j) piece(xs, wave * 8 + j, kt);
      }
    };
    if (nxt.valid) stage(0, 0);

    if (cur.valid) {   // epilogue of the previous tile, staged through ring buffer 1
      float bv[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int n = cur.n0 + wn * 64 + i * 16 + fg * 4 + r;
          bv[i][r] = (bias && n < N) ? bias[(int64_t)cur.e * N + n] : 0.f;
        }
#define OUT_ROUND(RAW, R, A, D, DST) \
  nt256p_out_round<TO, RAW, R, A, D>(acc, bv, DST, mul_pre, stg, cur, N, act, drop_p, seed, keep_scale, thresh16, tid, wm, wn, frow, fg)
#define OUT_PASS(A, D) { OUT_ROUND(false, 0, A, D, C); OUT_ROUND(false, 1, A, D, C); }
      if (pre_act) { OUT_ROUND(true, 0, APERTIS_ACT_NONE, false, pre_act); OUT_ROUND(true, 1, APERTIS_ACT_NONE, false, pre_act); }
      if (mul_pre) OUT_PASS(-1, false)
      else if (act == APERTIS_ACT_NONE && drop_p <= 0.f) OUT_PASS(APERTIS_ACT_NONE, false)
      else if (act == APERTIS_ACT_GELU && drop_p > 0.f) OUT_PASS(APERTIS_ACT_GELU, true)
      else if (act == APERTIS_ACT_GELU) OUT_PASS(APERTIS_ACT_GELU, false)
      else OUT_PASS(-1, false)
#undef OUT_PASS
#undef OUT_ROUND
    }
    if (!nxt.valid) break;
    cur = nxt;
    if (queue && tid == 0) *s_next = ntq_next(queue, (int)(blockIdx.x & 7), G, n_valid);   // (its latency hides under the K loop)

#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; ++kt) {
      // the wave's share of step kt's DMA has landed; a store wave has none while kt <= solo and
      // must not wait there (its output stores are still draining)
      if (wave < 4 || kt > solo) wait_vmcnt<0>();
      __syncthreads();
      if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
      const char *xs = smem + (kt & 1) * 2 * TILE2_BYTES, *ws = xs + TILE2_BYTES;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        frag wf[4], xf[8];
        const int chunk = kk * 4 + fg;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int wrow = wn * 64 + i * 16 + frow;
          wf[i] = *reinterpret_cast<const frag *>(ws + wrow * ROWB + ((chunk ^ (wrow & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          int xrow = wm * 128 + j * 16 + frow;
          xf[j] = *reinterpret_cast<const frag *>(xs + xrow * ROWB + ((chunk ^ (xrow & 7)) << 4));
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) mma(acc[i][j], wf[i], xf[j]);
        __builtin_amdgcn_s_setprio(0);
      }
    }
    __syncthreads();   // every wave is done with both ring buffers
    t = queue ? __builtin_amdgcn_readfirstlane(*s_next) : t + G;
  }
}

// ------------------------------------------------------------------------------------------
// Persistent 256 x 352 NT kernel for outputs whose width is a multiple of 352 (the H = 704 family: fc2 forward and the fc1
// data gradient at N = 704, the SSM input projection at N = 352).  On 256-wide tiles N = 704 is 256 + 256 + 192: the X
// panel enters LDS three times and a quarter of the third tile's MFMAs multiply zero columns; two 352-wide tiles read X
// twice and waste nothing - 17 % less L2 -> LDS fill (what bounds these GEMMs) and 8 % fewer MFMA slots
// per output.  Same ring, DMA, store-wave and queue scheme as grouped_gemm_nt256p_k; what differs: 8 waves as 4 (M) x 2 (N),
// wave tile 64 x 176 = 4 x 11 MFMA tiles (176 accumulator registers), a 76 KiB stage (32 KiB of X + 44 KiB of W), and the
// epilogue in four rounds of 64 rows (one m-subtile of every wave; 768-byte staging rows, chunk XOR row as before).
// Plain epilogue only (bias, conversion): the activation / dropout / second-output forms stay on the kernels above.
// ------------------------------------------------------------------------------------------
constexpr int BN5 = 352, W5_BYTES = BN5 * ROWB, BUF5 = TILE2_BYTES + W5_BYTES;   // 44 KiB of W, 76 KiB per stage
constexpr int STG5_PITCH = 768;                                                   // 44 chunks of 16 B + 4 of slack for the swizzle

template <typename TO>
__global__ void __launch_bounds__(NT2)
grouped_gemm_nt352p_k(const bf16_t *__restrict__ X, const bf16_t *__restrict__ W, const float *__restrict__ bias,
                      const int32_t *__restrict__ offsets, TO *__restrict__ C, int N, int K, int E, int n_tiles,
                      int total_tiles, int solo, int *__restrict__ queue) {
  typedef bf16_t T;
  typedef bf16x8 frag;
  static_assert(sizeof(TO) == 2, "staging layout assumes 2-byte outputs");
  constexpr int BK = 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *stg = smem + BUF5;                             // ring buffer 1 doubles as the C staging area
  int32_t *s_off = reinterpret_cast<int32_t *>(smem + 2 * BUF5);

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: the DMA piece arithmetic stays scalar)
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 15, fg = lane >> 4;
  const int G = gridDim.x;
  for (int i = tid; i <= E; i += NT2) s_off[i] = offsets[i];
  __syncthreads();

  int mt_valid = 0;
  for (int e = 0; e < E; ++e) mt_valid += (s_off[e + 1] - s_off[e] + BM2 - 1) / BM2;
  const int n_valid = __builtin_amdgcn_readfirstlane(min(total_tiles, mt_valid * n_tiles));
  const int nk = K / BK;   // K % 64 == 0 (launcher)

  f32x4 acc[11][4];
  PTile cur; cur.valid = 0; cur.e = 0; cur.rows_valid = 0; cur.n0 = 0; cur.cols_valid = 0; cur.row0 = 0;
  int *s_next = s_off + 1026;
  for (int t = blockIdx.x;;) {
    PTile nxt; nxt.valid = 0;
    if (t < n_valid) {
      const int r = t / G, within = t - r * G, gr = min(G, n_valid - r * G);
      const int tile = r * G + xcd_remap(within, gr);
      const int mt = tile / n_tiles, ntile = tile - mt * n_tiles;
      int accm = 0;
      for (int e = 0; e < E; ++e) {
        const int r0 = s_off[e], r1 = s_off[e + 1];
        const int nt = (r1 - r0 + BM2 - 1) / BM2;
        if (mt < accm + nt) {
          const int m0 = (mt - accm) * BM2;
          nxt.valid = 1;
          nxt.e = __builtin_amdgcn_readfirstlane(e);
          nxt.row0 = (int64_t)__builtin_amdgcn_readfirstlane(r0 + m0);
          nxt.rows_valid = __builtin_amdgcn_readfirstlane(min(BM2, r1 - r0 - m0));
          nxt.n0 = __builtin_amdgcn_readfirstlane(ntile * BN5);
          nxt.cols_valid = __builtin_amdgcn_readfirstlane(min(BN5, N - ntile * BN5));
          break;
        }
        accm += nt;
      }
    }
    const int ldb = K * (int)sizeof(T);   // both operands: rows of K elements (ldw == K, launcher)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T *>(X + nxt.row0 * K), 0, nxt.rows_valid * ldb, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T *>(W + ((int64_t)nxt.e * N + nxt.n0) * K), 0, nxt.cols_valid * ldb, 0x00020000);
    const int voff0 = (lane >> 3) * ldb + (((lane & 7) ^ (lane >> 3)) << 4);
    // a run of `cnt` consecutive pieces (8 rows x 128 B each) of one operand from piece p0 on; the lane offset (which carries the
    // piece's row term - the hardware range check covers nothing else) is a running sum, pinned so that hipcc does not keep a
    // table of all of a wave's offsets alive across the K loop (it spilled them)
    auto run = [&](const __amdgpu_buffer_rsrc_t &rs, char *dst, int p0, int cnt, int kt) {
      int v = voff0 + p0 * 8 * ldb;
      asm volatile("" : "+v"(v));
      char *d = dst + p0 * 1024;
#pragma unroll
      for (int j = 0; j < cnt; ++j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(d + j * 1024), 16, v,
                                                 kt * BK * (int)sizeof(T), 0, 0);
        v += 8 * ldb;
      }
    };
    // K step kt's operands (32 X pieces + 44 W pieces): by waves 0-3 alone while kt <= solo
    auto stage = [&](int buf, int kt) {
      char *xs = smem + buf * BUF5;
      if (kt > solo) {
        run(xrs, xs, wave * 4, 4, kt);
        if (wave < 4) run(wrs, xs + TILE2_BYTES, wave * 6, 6, kt);
        else run(wrs, xs + TILE2_BYTES, 24 + (wave - 4) * 5, 5, kt);
      } else if (wave < 4) {
        run(xrs, xs, wave * 8, 8, kt);
        run(wrs, xs + TILE2_BYTES, wave * 11, 11, kt);
      }
    };
    if (nxt.valid) stage(0, 0);

    if (cur.valid) {   // epilogue of the previous tile: four rounds of 64 rows through ring buffer 1
      int frow_e = frow, tid_e = tid;   // (pinned: the staging addresses are not to live across the K loop)
      asm volatile("" : "+v"(frow_e), "+v"(tid_e));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < 11; ++i) {
          const int ncol = wn * 176 + i * 16 + fg * 4;            // column inside the tile (N % 4 == 0: valid in fours)
          float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
          if (bias && ncol < cur.cols_valid) b = *reinterpret_cast<const float4 *>(bias + (int64_t)cur.e * N + cur.n0 + ncol);
          const float bq[4] = {b.x, b.y, b.z, b.w};
          float o[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float a = acc[i][r][q];
            asm volatile("" : "+v"(a));   // pins the conversion inside its round
            o[q] = a + bq[q];
          }
          const int srow = wm * 16 + frow_e, chunk = wn * 22 + i * 2 + (fg >> 1);
          *reinterpret_cast<uint2 *>(stg + srow * STG5_PITCH + ((chunk ^ frow_e) << 4) + (fg & 1) * 8) =
              make_uint2(pack_bf16x2_any(o[0], o[1]), pack_bf16x2_any(o[2], o[3]));
        }
        __syncthreads();
        if (tid >= NT2 / 2) {   // store waves: 64 rows x 44 chunks
#pragma unroll
          for (int it = 0; it < 11; ++it) {
            const int c4 = it * (NT2 / 2) + tid_e - NT2 / 2;
            const int srow = c4 / 44, c = c4 - srow * 44;
            const int row = (srow >> 4) * 64 + r * 16 + (srow & 15);
            if (row < cur.rows_valid && c * 8 < cur.cols_valid) {
              const uint4 v = *reinterpret_cast<const uint4 *>(stg + srow * STG5_PITCH + ((c ^ (srow & 15)) << 4));
              out_store16(C + (cur.row0 + row) * N + cur.n0 + c * 8, v);
            }
          }
        }
        __syncthreads();
      }
    }
    if (!nxt.valid) break;
    cur = nxt;
    if (queue && tid == 0) *s_next = ntq_next(queue, (int)(blockIdx.x & 7), G, n_valid);

#pragma unroll
    for (int i = 0; i < 11; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // (the tile's fields went through the loop-carried `cur`: hipcc no longer knows that they are wave-uniform and would
    // build every piece's descriptor in VGPRs behind a waterfall loop)
    auto uni_ptr = [](const T *q) {
      const uint64_t v = (uint64_t)q;
      return (const T *)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                         (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v));
    };
    const int pw0 = wave < 4 ? wave * 6 : 24 + (wave - 4) * 5;     // this wave's first W piece (the split of `stage`)
    const int rows_w = __builtin_amdgcn_readfirstlane(cur.rows_valid) - wave * 32;
    const int cols_w = __builtin_amdgcn_readfirstlane(cur.cols_valid) - pw0 * 8;
    const T *fxw = uni_ptr(X + (cur.row0 + wave * 32) * K), *fww = uni_ptr(W + ((int64_t)cur.e * N + cur.n0 + pw0 * 8) * K);
    for (int kt = 0; kt < nk; ++kt) {
      if (wave < 4 || kt > solo) wait_vmcnt<0>();
      __syncthreads();
      // While waves 0-3 fill alone (the next tile's first `solo` steps) the next K step's pieces are issued here, as ever.
      // In the steady state they are NOT issued in one run: a wave sits in each `buffer_load ... lds` until the CU's
      // address unit has taken it (16 cycles per KiB piece; 76 pieces per step and CU, all eight waves at once right
      // behind the barrier = ~1200 cycles in which no MFMA issues - the 0.4-0.5 us that round 2's probes saw the
      // "asynchronous" fills add to every step).  They go out one at a time behind every second group of four MFMAs
      // below: this wave's four X pieces, then its six W pieces (waves 4-7 have five).
      const int nkt = kt + 1;
      const bool spread = nkt < nk && nkt > solo;
      if (nkt < nk && nkt <= solo) stage(nkt & 1, nkt);
      char *fxs = smem + (nkt & 1) * BUF5 + wave * 4 * 1024, *fws = smem + (nkt & 1) * BUF5 + TILE2_BYTES + pw0 * 1024;
      // a piece's row term goes into its buffer descriptor (base advanced and range shortened by 8 rows per piece: scalar
      // arithmetic), so that every piece uses the SAME lane offset voff0 - a running per-lane offset cost two registers the
      // kernel does not have (hipcc spilled it around every DMA)
      auto fill_piece = [&](int q) {   // q = 0..9, a compile-time constant at every call site
        if (q < 4) {
          const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
              const_cast<T *>(fxw + q * 8 * K), 0, max(rows_w - q * 8, 0) * ldb, 0x00020000);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(fxs + q * 1024), 16, voff0,
                                                   nkt * BK * (int)sizeof(T), 0, 0);
        } else {
          // (waves 4-7 have five W pieces: their sixth slot repeats the fifth - the same bytes to the same place; a piece
          // from beyond the range would still WRITE its zeros, over the next wave's first piece)
          const int j = (q == 9 && wave >= 4) ? 4 : q - 4;
          const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
              const_cast<T *>(fww + j * 8 * K), 0, max(cols_w - j * 8, 0) * ldb, 0x00020000);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(fws + j * 1024), 16, voff0,
                                                   nkt * BK * (int)sizeof(T), 0, 0);
        }
      };
      const char *xs = smem + (kt & 1) * BUF5, *ws = xs + TILE2_BYTES;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        // 176 accumulator registers leave room for the four X fragments and a few W fragments, not for all eleven: the W
        // fragments are read three ahead of the MFMAs that consume them
        frag xf[4], wf[11];
        const int chunk = kk * 4 + fg;
        auto wread = [&](int i) {
          const int wrow = wn * 176 + i * 16 + frow;
          return *reinterpret_cast<const frag *>(ws + wrow * ROWB + ((chunk ^ (wrow & 7)) << 4));
        };
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int xrow = wm * 64 + j * 16 + frow;
          xf[j] = *reinterpret_cast<const frag *>(xs + xrow * ROWB + ((chunk ^ (xrow & 7)) << 4));
        }
        wf[0] = wread(0); wf[1] = wread(1); wf[2] = wread(2);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 11; ++i) {
          if (i + 3 < 11) wf[i + 3] = wread(i + 3);
          __builtin_amdgcn_sched_barrier(0);   // keeps the reads from being hoisted into one block of 44 live registers
#pragma unroll
          for (int j = 0; j < 4; ++j) mma(acc[i][j], wf[i], xf[j]);
          const int g = kk * 11 + i;           // MFMA group 0..21 of the step
          if (!(g & 1) && g < 20) {
            __builtin_amdgcn_sched_barrier(0);
            if (spread) fill_piece(g >> 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __builtin_amdgcn_s_setprio(0);
      }
    }
    __syncthreads();   // every wave is done with both ring buffers
    t = queue ? __builtin_amdgcn_readfirstlane(*s_next) : t + G;
  }
}

// ------------------------------------------------------------------------------------------
// 256 x 128 NT kernel, TWO work-groups per CU.  The persistent kernel above runs a tile's epilogue
// (VALU: activation, dropout hash, conversions; then the output stores) with the matrix pipes idle and
// its K loop with the VALU idle, and keeps at most one 64 KiB stage of operands in flight per CU.  Here
// a work-group is 4 waves (one per SIMD; wave tile 64 rows x 128 columns, the ACTIVATION rows on the MFMA A
// operand - nt2x_epilogue says why) with its own 72 KiB ring of three
// 32-deep stages, so two independent work-groups share a CU: one's epilogue runs under the other's
// MFMAs, and 2 x 48 KiB of LDS-DMA are in flight per CU.  Inside a wave the fragments of sub-step s+1
// are read from LDS under the MFMAs of sub-step s (W fragments in place once their four MFMAs have
// issued, X fragments into a second set).  Rows are 64 B per stage; the 16-byte chunk c of row r sits
// at chunk position c ^ F[(r >> 2) & 3], F = {0,3,2,1}, which makes both the 1 KiB DMA pieces (16 rows)
// and the ds_read_b128 fragment reads (lane groups of MI355X_MICROARCH.md's LDS table) conflict-free.
// ------------------------------------------------------------------------------------------
constexpr int NT3 = 256, BM3 = 256, BN3 = 128, ROWB3 = 64;
constexpr int SLOT3 = (BM3 + BN3) * ROWB3;   // 24 KiB: X rows, then W rows
constexpr int RING3 = 3 * SLOT3;

__device__ __forceinline__ void lds_dma16s(const v4i &rs, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(__builtin_amdgcn_readfirstlane((int)lds_addr)), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}

// eight bf16 products (fp32 multiply, rounded once)
__device__ __forceinline__ uint4 mul_chunk_bf16(uint4 a, uint4 b) {
  const uint32_t x[4] = {a.x, a.y, a.z, a.w}, y[4] = {b.x, b.y, b.z, b.w};
  uint32_t o[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const float p0 = __builtin_bit_cast(float, x[w] << 16) * __builtin_bit_cast(float, y[w] << 16);
    const float p1 = __builtin_bit_cast(float, x[w] & 0xffff0000u) * __builtin_bit_cast(float, y[w] & 0xffff0000u);
    o[w] = pack_bf16x2_any(p0, p1);
  }
  return make_uint4(o[0], o[1], o[2], o[3]);
}

// Helpers of the saved-gradient epilogue below (round 3: its VALU work - 65 instructions per element pair, a sixth of them
// bit fiddling around the mask and the conversions - is what the fc1 forward spends its time on next to the MFMAs).
// Two floats -> one dword of bf16 (one v_cvt_pk_bf16_f32; a conversion per element wastes half of each).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  typedef float f2_t __attribute__((ext_vector_type(2)));
  const f2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// 0xFFFF in each DROPPED half of a hash word: keep <=> r16 >= thresh16 (drop_keep4's rule) <=> clamp(thresh - r, 0) == 0,
// on the packed 16-bit pipe (three instructions per two elements instead of two extractions, two compares and a select
// per output)
// (as inline assembly: from the vector builtins hipcc makes two 16-bit compares, two selects and a byte permute)
__device__ __forceinline__ uint32_t drop_mask2(uint32_t h, uint32_t thresh16) {
  const uint32_t t2 = __builtin_amdgcn_readfirstlane((int)(thresh16 | (thresh16 << 16)));
  uint32_t d;
  asm("v_pk_sub_u16 %0, %1, %2 clamp\n\tv_pk_min_u16 %0, %0, %3\n\tv_pk_mul_lo_u16 %0, %0, %4"
      : "=&v"(d) : "s"(t2), "v"(h), "s"(0x00010001u), "s"(0xffffffffu));
  return d;
}

// Epilogue of the two-per-CU kernel, STRAIGHT FROM THE ACCUMULATORS (round 4).  In this kernel the ACTIVATION tile feeds the
// MFMA A operand and the WEIGHT tile the B operand (the reverse of the other kernels here), the four waves split the tile's
// rows (wave tile 64 x 128), and W's rows are permuted on their way into LDS (LDS row j*16 + c of the tile holds W row c*8 + j).
// Lane (c = lane & 15, g = lane >> 4) then holds, for each of its 16 rows m = wave*64 + i*16 + g*4 + q, the EIGHT consecutive
// columns n = c*8 + j (j = 0..7) - one 16-byte piece of a bf16 output row - and the 16 lanes of a row group cover the row's 256
// bytes: every wave-instruction stores (or, for the saved tensor, loads) four whole rows of the tile.  No LDS staging pass, no
// barrier: a wave's first stores leave while it evaluates its later rows, and no wave waits for another in the epilogue.
// (Round 2 measured a direct form on the OLD operand roles - lane = row, 32-byte pieces, 16 rows per instruction - at the
// staged form's speed; rounds 2-3: with the 8-byte lane pieces of the accumulators as they stood, partial-line stores lost.)
// Arithmetic and rounding are those of the staged epilogues this replaces, element for element (outputs bit-identical).
//   EPI_RAW      dst  = acc + bias                                   (the pre-activation)
//   EPI_ACT      dst  = dropout(act(bf16(acc + bias)))
//   EPI_BOTH     dst  = gelu(pre) * mask / (1-p), dst2 = gelu'(pre) * mask / (1-p)   (APERTIS_ACT_SAVE_GRAD: ONE evaluation and
//                ONE hash per element yield both; gelu and gelu' share their folded erfc terms)
//   EPI_MULACT   dst  = bf16(acc) * act'(saved pre) * mask / (1-p)   (fused data gradient, pre-activation form)
//   EPI_MULSAVED dst  = bf16(acc) * saved g'                         (APERTIS_ACT_MUL_SAVED)
enum { EPI_RAW = 0, EPI_ACT = 1, EPI_BOTH = 2, EPI_MULACT = 3, EPI_MULSAVED = 4 };

// HASB = false: no bias term (the data-gradient forms of the ring kernel, whose launcher refuses a bias beside mul_pre): the
// add of eight zeros cost the row piece a v_pk_add_f32 per pair AND two v_mov_b32 per pair that bring the addends - which sit
// in eight different accumulator quads - into adjacent registers: 384 of the ~1 000 VALU instructions of the saved-gradient
// multiply's epilogue.  (acc + 0.0f == acc bit for bit: an accumulator that starts at +0 is never -0.)
template <typename TO, int MODE, int ACT, bool DROP, bool HASB = true>
__device__ __forceinline__ void nt2x_epilogue(const f32x4 (&acc)[4][8], const float (&bv)[8], TO *__restrict__ dst,
                                              TO *__restrict__ dst2, const TO *__restrict__ saved, int64_t row0, int rows_valid,
                                              int n0, int cols_valid, int N, int act, float drop_p, uint64_t seed,
                                              float keep_scale, uint32_t thresh16, int wave, int frow, int fg) {
  static_assert(sizeof(TO) == 2, "16-byte pieces of 2-byte outputs");
  typedef unsigned u4_t __attribute__((ext_vector_type(4)));
  const int64_t tile0 = row0 * N + n0;                    // wave-uniform: the tile's first element
  const int rl = wave * 64 + fg * 4;                      // this lane's first tile row (+ i*16 + q)
  // Raw buffer descriptors over the tile's valid rows: the hardware drops a store (returns zeros for a load) whose lane offset
  // lies past them, so ragged tiles need neither branches nor exec masks around the 32 stores; lanes past the last valid column
  // of a partial n-tile start from an offset that no row term brings back into range (a tile spans 256 * N * 2 bytes << 2^30).
  // (The range check covers the lane offset only: the row term is added to it, not passed as the scalar offset.)
  const uint32_t voff = frow * 8 < cols_valid ? ((uint32_t)rl * (uint32_t)N + (uint32_t)(frow * 8)) * 2u : 0xC0000000u;
  const uint32_t tile_bytes = (uint32_t)rows_valid * (uint32_t)N * 2u;
  const __amdgpu_buffer_rsrc_t o1 = __builtin_amdgcn_make_buffer_rsrc(dst + tile0, 0, tile_bytes, 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t o2 = __builtin_amdgcn_make_buffer_rsrc(MODE == EPI_BOTH ? dst2 + tile0 : dst + tile0, 0, tile_bytes, 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t sv = __builtin_amdgcn_make_buffer_rsrc(const_cast<TO *>(MODE >= EPI_MULACT ? saved + tile0 : dst + tile0), 0, tile_bytes, 0x00020000);
  auto row_off = [&](int r) { return voff + (uint32_t)(r * N) * 2u; };
  // (non-temporal, like out_store16: the outputs are far larger than the caches and would only displace the operands)
  auto put = [&](const __amdgpu_buffer_rsrc_t &rs, uint32_t off, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    __builtin_amdgcn_raw_buffer_store_b128((u4_t){a, b, c, d}, rs, (int)off, 0, 2);
  };
  // the saved tensor's pieces: two batches of four rows in flight ahead of the arithmetic
  [[maybe_unused]] uint4 pc[2][4];
  [[maybe_unused]] auto fetch = [&](int i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const u4_t t = __builtin_amdgcn_raw_buffer_load_b128(sv, (int)row_off(i * 16 + q), 0, 0);
      pc[i & 1][q] = make_uint4(t[0], t[1], t[2], t[3]);
    }
  };
  if constexpr (MODE >= EPI_MULACT) {
    fetch(0);
    fetch(1);
    __builtin_amdgcn_sched_barrier(0);
  }
  // GELU constants with the dropout's 1 / (1 - p) folded in (gelu_consts), and the column halves of the lane's four mask
  // hashes (gd_colmix: once per tile; a row then costs one gd_rowmix and its pairs one xor-multiply-xorshift each)
  [[maybe_unused]] const GeluK gk = gelu_consts(keep_scale);
  [[maybe_unused]] uint32_t cmix[4] = {0u, 0u, 0u, 0u};
  // ... and the row halves: the 16 lanes of a row group would each run the full finaliser for each of their 16 rows (144 of the
  // row pieces' 2 441 VALU instructions per tile); lane l computes the one of the wave's row l instead and a row's lanes fetch
  // it through the LDS crossbar (ds_bpermute: no VALU slot, no LDS memory)
  [[maybe_unused]] int rmix_mine = 0;
  if constexpr (MODE == EPI_BOTH && DROP) {
#pragma unroll
    for (int w = 0; w < 4; ++w) cmix[w] = gd_colmix(seed, (uint32_t)((n0 + frow * 8) >> 1) + (uint32_t)w);
    rmix_mine = (int)gd_rowmix(seed, (uint64_t)(row0 + wave * 64 + (frow + fg * 16)));
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = i * 16 + q;
      const uint32_t off = row_off(r);
      float v[8];
      if constexpr (MODE == EPI_BOTH) {
        // (one v_add_f32 per element: the eight addends of a row sit in eight different accumulator quads, and for a
        // v_pk_add_f32 hipcc first copies each pair into adjacent registers - three instructions for two sums)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm("v_add_f32 %0, %1, %2" : "=v"(v[j]) : "v"(acc[i][j][q]), "v"(bv[j]));
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = HASB ? acc[i][j][q] + bv[j] : acc[i][j][q];
      }
      uint32_t o[4];
      if constexpr (MODE == EPI_BOTH) {
        // mask words: 0xFFFF where the element is dropped (one hash word per pair of elements)
        uint32_t og[4], dm[4] = {0u, 0u, 0u, 0u};
        if (DROP) {
          const uint32_t rmix = (uint32_t)__builtin_amdgcn_ds_bpermute((fg * 4 + r) * 4, rmix_mine);   // row rl + r of the tile
#pragma unroll
          for (int w = 0; w < 4; ++w) dm[w] = gd_pair(rmix, cmix[w]);
#pragma unroll
          for (int w = 0; w < 4; ++w) dm[w] = drop_mask2(dm[w], thresh16);
        }
        v2f x[4], hv[4], gv[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {   // the pre-activation as the other form stores it (rounded to bf16)
          const uint32_t xpk = pack_bf16x2(v[2 * w], v[2 * w + 1]);
          x[w] = (v2f){__builtin_bit_cast(float, xpk << 16), __builtin_bit_cast(float, xpk & 0xffff0000u)};
        }
        gelu_both_fast8(x, gk, hv, gv);   // (scaled by 1 / (1 - p) already)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          o[w] = pack_bf16x2(hv[w].x, hv[w].y) & ~dm[w];
          og[w] = pack_bf16x2(gv[w].x, gv[w].y) & ~dm[w];
        }
        put(o1, off, o[0], o[1], o[2], o[3]);
        put(o2, off, og[0], og[1], og[2], og[3]);
      } else {
        if constexpr (MODE == EPI_ACT) {
          bool keep[8] = {true, true, true, true, true, true, true, true};
          if (ACT >= 0 ? DROP : drop_p > 0.f) {
            bool k4[4];
            gd_keep4(seed, row0 + rl + r, n0 + frow * 8, thresh16, k4);
#pragma unroll
            for (int j = 0; j < 4; ++j) keep[j] = k4[j];
            gd_keep4(seed, row0 + rl + r, n0 + frow * 8 + 4, thresh16, k4);
#pragma unroll
            for (int j = 0; j < 4; ++j) keep[4 + j] = k4[j];
          }
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if constexpr (ACT == APERTIS_ACT_GELU) {   // the scale folded in: the same operations as EPI_BOTH's, bit for bit
              v[j] = gelu_fast(to_f32(from_f32<TO>(v[j])), gk);
              v[j] = keep[j] ? v[j] : 0.f;
            } else {
              if (ACT != APERTIS_ACT_NONE) v[j] = act_fwd<true>(to_f32(from_f32<TO>(v[j])), ACT >= 0 ? ACT : act);
              v[j] = keep[j] ? v[j] * keep_scale : 0.f;   // keep_scale is 1 without dropout
            }
          }
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) o[w] = pack_bf16x2(v[2 * w], v[2 * w + 1]);
        uint4 ov = make_uint4(o[0], o[1], o[2], o[3]);
        if constexpr (MODE == EPI_MULSAVED) ov = mul_chunk_bf16(ov, pc[i & 1][q]);
        if constexpr (MODE == EPI_MULACT)
          ov = actbwd_chunk<TO, true, ACT, DROP>(ov, pc[i & 1][q], row0 + rl + r, n0 + frow * 8, N, act, drop_p, seed, keep_scale, thresh16);
        put(o1, off, ov.x, ov.y, ov.z, ov.w);
      }
    }
    if constexpr (MODE >= EPI_MULACT) {
      __builtin_amdgcn_sched_barrier(0);
      if (i + 2 < 4) fetch(i + 2);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// RAGGED: K % 32 != 0 - W's rows are zero-padded to whole sub-steps (row pitch ldw), X's last sub-step over-reads into
// the next row (finite values against W's zeros) and the K offset moves into the range-checked lane offset so that
// the tile's last row reads zeros past the buffer instead
template <typename TO, bool RAGGED = false>
__global__ void __launch_bounds__(NT3, 2)
grouped_gemm_nt2x_k(const bf16_t *__restrict__ X, const bf16_t *__restrict__ W, const float *__restrict__ bias,
                    const int32_t *__restrict__ offsets, TO *__restrict__ C, TO *__restrict__ pre_act,
                    const TO *__restrict__ mul_pre, int N, int K, int ldw, int E, int n_tiles, int act_flags, float drop_p,
                    uint64_t seed, int walk_g, int walk_nb, int spread_fill) {
  typedef bf16x8 frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // act_flags: the activation code, | APERTIS_ACT_SAVE_GRAD (forward, GELU: pre_act receives gelu'(pre) * mask / (1-p) instead
  // of pre) or == APERTIS_ACT_MUL_SAVED (data gradient: the output is multiplied by that saved tensor, passed as mul_pre)
  const bool save_grad = (act_flags & APERTIS_ACT_SAVE_GRAD) != 0, mul_saved = (act_flags & APERTIS_ACT_MUL_SAVED) != 0;
  const int act = act_flags & 0xff;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fg = lane >> 4;

  // tile of this work-group: an XCD takes a contiguous run of the tile order (xcd_remap); inside it the walk is
  // n-panel-stationary (tile_walk below) so that the W panel the XCD's resident work-groups share stays in its L2
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  int mt, ntile;
  tile_walk(tile, gridDim.x / n_tiles, n_tiles, walk_g, walk_nb, mt, ntile);
  int e = -1, rows_valid = 0, accm = 0;
  int64_t row0 = 0;
  for (int g = 0; g < E; ++g) {
    const int r0 = offsets[g], r1 = offsets[g + 1];
    const int nt = (r1 - r0 + BM3 - 1) / BM3;
    if (mt < accm + nt) {
      const int m0 = (mt - accm) * BM3;
      e = g; row0 = r0 + m0; rows_valid = min(BM3, r1 - r0 - m0);
      break;
    }
    accm += nt;
  }
  if (e < 0) return;
  e = __builtin_amdgcn_readfirstlane(e);
  rows_valid = __builtin_amdgcn_readfirstlane(rows_valid);
  row0 = (int64_t)__builtin_amdgcn_readfirstlane((int)row0);
  const int n0 = ntile * BN3, cols_valid = min(BN3, N - n0);

  // operands through raw buffer descriptors sized to the tile's valid rows (zero fill past them); a lane's
  // offset = row-in-piece and swizzled chunk, the K offset of the sub-step rides in the scalar offset
  const int ldb = K * 2, ldwb = ldw * 2;
  const v4i xrs = raw_buffer_rsrc(X + row0 * K, (uint32_t)rows_valid * (uint32_t)ldb);
  const v4i wrs = raw_buffer_rsrc(W + ((int64_t)e * N + n0) * ldw, (uint32_t)cols_valid * (uint32_t)ldwb);
  const int fsw = (4 - ((lane >> 4) & 3)) & 3;                       // F[(row >> 2) & 3] for row = lane >> 2
  const uint32_t vx0 = (uint32_t)((wave * 64 + (lane >> 2)) * ldb + (((lane & 3) ^ fsw) << 4));
  // W's rows are permuted on the way in: LDS row j*16 + c of the tile holds W row c*8 + j (nt2x_epilogue: a lane's eight
  // accumulator columns are then eight consecutive output columns); piece p = rows p*16 + (lane >> 2), so + p rows per piece
  const uint32_t vw0 = (uint32_t)((lane >> 2) * 8 * ldwb + (((lane & 3) ^ fsw) << 4));
  const uint32_t lds0 = lds_addr_of(smem);
  // piece q of this wave's share of sub-step s: q = 0..3 its X pieces, 4..5 its W pieces
  auto issue_piece = [&](uint32_t slot_off, int s, int q) {
    const uint32_t kb = (uint32_t)s * ROWB3, base = lds0 + slot_off;
    const uint32_t kv = RAGGED ? kb : 0u, ks = RAGGED ? 0u : kb;
    if (q < 4) lds_dma16s(xrs, base + (wave * 4 + q) * 1024, vx0 + (uint32_t)(q * 16 * ldb) + kv, ks);
    else lds_dma16s(wrs, base + BM3 * ROWB3 + (wave * 2 + (q - 4)) * 1024, vw0 + (uint32_t)((wave * 2 + (q - 4)) * ldwb) + kv, ks);
  };
  auto issue = [&](uint32_t slot_off, int s) {   // sub-step s: this wave's 4 X pieces and 2 W pieces
#pragma unroll
    for (int q = 0; q < 6; ++q) issue_piece(slot_off, s, q);
  };
  const int nk = (K + 31) / 32;   // >= 3 (launcher)
  (void)spread_fill;
  issue(0, 0); issue(SLOT3, 1); issue(2 * SLOT3, 2);

  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment addresses: the lane part is the same for every 16-row sub-tile.  A operand = this wave's 64 activation rows
  // (4 fragments, double-buffered across sub-steps), B operand = the tile's 128 (permuted) weight rows (8 fragments, refilled
  // in place once their four MFMAs have issued): acc[i][j] = rows wave*64 + i*16 + (fg*4 + q), column-index frow of W sub-tile j
  const int frd = frow * ROWB3 + ((fg ^ ((4 - ((frow >> 2) & 3)) & 3)) << 4);
  const char *abase = smem + wave * 64 * ROWB3 + frd, *bbase = smem + BM3 * ROWB3 + frd;
  frag af[2][4], bfr[8];
  auto load_a = [&](frag (&dst)[4], int slot_off) {
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = *reinterpret_cast<const frag *>(abase + slot_off + i * 16 * ROWB3);
  };
  // 32 MFMAs on (acur, bfr) while (anxt, bfr) are refilled from the slot at nxt_off
  auto sub_step = [&](const frag (&acur)[4], frag (&anxt)[4], int nxt_off) {
    load_a(anxt, nxt_off);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int i = 0; i < 4; ++i) mma(acc[i][j], acur[i], bfr[j]);
      bfr[j] = *reinterpret_cast<const frag *>(bbase + nxt_off + j * 16 * ROWB3);
    }
    // issue order: 4 MFMAs, then the LDS reads whose registers they freed (hipcc otherwise sinks the reads to
    // the end of the MFMA run, where their latency is exposed)
#define SGB(nr) __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, nr, 0);
    SGB(3) SGB(3) SGB(1) SGB(1) SGB(1) SGB(1) SGB(1) SGB(1)
#undef SGB
  };
  // (round 3, measured at N = 2816, K = 704 and removed: the DMA pieces of a sub-step going out one by one behind the first six
  // MFMA groups instead of in one run behind the barrier - 1486 vs 1491 us, 1244 vs 1249: this kernel's K loop is not bound there)
#define NT2X_STEP(S, AC, AN)                                                                             \
      if ((S) + 3 < nk) issue((uint32_t)cur, (S) + 3);                                                   \
      __builtin_amdgcn_s_setprio(1);                                                                     \
      sub_step(AC, AN, nxt);   /* (past the last sub-step: harmless reads of a stale slot) */            \
      __builtin_amdgcn_s_setprio(0);
  wait_vmcnt<12>();   // stage 0 (vmcnt retires in order)
  lds_barrier();
  load_a(af[0], 0);
#pragma unroll
  for (int j = 0; j < 8; ++j) bfr[j] = *reinterpret_cast<const frag *>(bbase + j * 16 * ROWB3);
  int cur = 0;   // LDS offset of sub-step s's slot
  for (int s = 0; s < nk; s += 2) {
    // top of a sub-step: this wave holds the fragments of s (slot s is free once every wave says so) and its
    // share of stage s+1 has landed; behind the barrier stage s+1 is complete and stage s+3 may overwrite s
#define SUB(S, AC, AN)                                                                                   \
    {                                                                                                    \
      const int nxt = cur + SLOT3 == RING3 ? 0 : cur + SLOT3;                                            \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
      if ((S) + 2 < nk) wait_vmcnt<6>(); else wait_vmcnt<0>();                                           \
      lds_barrier();                                                                                   \
      NT2X_STEP(S, AC, AN)                                                                               \
      cur = nxt;                                                                                         \
    }
    SUB(s, af[0], af[1])
    if (s + 1 < nk) SUB(s + 1, af[1], af[0])
    else {   // odd nk: keep the register roles of the loop
#pragma unroll
      for (int i = 0; i < 4; ++i) af[0][i] = af[1][i];
    }
#undef SUB
#undef NT2X_STEP
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // (no barrier: the epilogue leaves straight from the accumulators, the ring is not touched again)

  const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const uint32_t thresh16 = (uint32_t)(drop_p * 65536.f);
  float bv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int n = n0 + frow * 8 + j;
    bv[j] = (bias && n < N) ? bias[(int64_t)e * N + n] : 0.f;
  }
#define OUT(MODE, A, D, DST, DST2) \
  nt2x_epilogue<TO, MODE, A, D>(acc, bv, DST, DST2, mul_pre, row0, rows_valid, n0, cols_valid, N, act, drop_p, seed, keep_scale, thresh16, wave, frow, fg)
  if (save_grad && pre_act) {   // (GELU: launch_nt refuses the flag for other activations)
    if (drop_p > 0.f) OUT(EPI_BOTH, APERTIS_ACT_GELU, true, C, pre_act);
    else OUT(EPI_BOTH, APERTIS_ACT_GELU, false, C, pre_act);
    return;
  }
  if (pre_act) OUT(EPI_RAW, APERTIS_ACT_NONE, false, pre_act, nullptr);
  if (mul_pre) {
    if (mul_saved) OUT(EPI_MULSAVED, APERTIS_ACT_NONE, false, C, nullptr);
    else if (act == APERTIS_ACT_GELU && drop_p > 0.f) OUT(EPI_MULACT, APERTIS_ACT_GELU, true, C, nullptr);
    else if (act == APERTIS_ACT_GELU) OUT(EPI_MULACT, APERTIS_ACT_GELU, false, C, nullptr);
    else OUT(EPI_MULACT, -1, false, C, nullptr);
  } else if (act == APERTIS_ACT_NONE && drop_p <= 0.f) OUT(EPI_RAW, APERTIS_ACT_NONE, false, C, nullptr);
  else if (act == APERTIS_ACT_GELU && drop_p > 0.f) OUT(EPI_ACT, APERTIS_ACT_GELU, true, C, nullptr);
  else if (act == APERTIS_ACT_GELU) OUT(EPI_ACT, APERTIS_ACT_GELU, false, C, nullptr);
  else OUT(EPI_ACT, -1, false, C, nullptr);
#undef OUT
}

// ------------------------------------------------------------------------------------------
// Skinny NT kernel (round 4): a handful of rows per call - the single-token decode step (reference core.py:1578-1603: B <= 16
// rows through every projection and through the experts a token chose).  The 128 x 128 kernel below spent 21.5 us per call
// there (one 128-row tile per group for 1-16 rows, N / 128 work-groups, operands through LDS): 5 calls per layer = over half
// of a captured token step.  Here a WAVE owns 16 output columns of one group: W rows stream straight from global memory
// into the MFMA A operand (16 B per lane and 32-deep step, eight steps in flight), the rows of X are the B operand, nothing
// goes through LDS, N / 16 waves per group fill the chip, and a lane ends up with four consecutive outputs of one row (one
// 8-byte store).  The k order (for K < 512) and the epilogue arithmetic are the 128 x 128 kernel's.  bf16, no dropout / second output.
// ------------------------------------------------------------------------------------------
// KS > 1 (long K: the experts' second GEMM, K = I): KS waves of a work-group split the K range, their partial sums meet in LDS
// and wave 0 adds them in wave order - a wave alone walked 88 dependent load -> MFMA batches' worth of K = 2816.
#ifndef SKINNY_LONG_K_WAVES
#define SKINNY_LONG_K_WAVES 16      // (probe switch: 4 = the K >= 512 form for every long K)
#endif
template <typename TO, int KS>
__global__ void __launch_bounds__(64 * KS)
grouped_gemm_nt_skinny_k(const bf16_t *__restrict__ X, const bf16_t *__restrict__ W, const float *__restrict__ bias,
                         const int32_t *__restrict__ offsets, TO *__restrict__ C, int N, int K, int ldw, int act, int max_rows) {
  typedef bf16x8 frag;
  constexpr int U = 8;
  __shared__ f32x4 s_part[KS > 1 ? KS - 1 : 1][64];
  const int e = blockIdx.y, n0 = blockIdx.x * 16, lane = threadIdx.x & 63, ks = threadIdx.x >> 6;
  const int kq = KS > 1 ? ((K + KS * 32 - 1) / (KS * 32)) * 32 : K;        // this wave's K range: [ks * kq, min(K, ks * kq + kq))
  const int kbeg = ks * kq, kend = min(K, kbeg + kq);
  const int r0 = offsets[e], rows = min(offsets[e + 1], max_rows) - r0;     // (rows >= max_rows are neither read nor written)
  if (rows <= 0) return;
  const int l15 = lane & 15, fg = lane >> 4, kc = fg * 8;
  const int wcol = n0 + l15;                                     // this lane's W row (= output column) as the A operand's row
  const bf16_t *wrow = W + ((int64_t)e * N + min(wcol, N - 1)) * ldw;
  const bool w_ok = wcol < N;
  const int nq = n0 + fg * 4;                                    // the four output columns this lane ends up with
  float bq[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[q] = nq + q < N ? bias[(int64_t)e * N + nq + q] : 0.f;
  }
  const frag zero = {};
  for (int rb = 0; rb < rows; rb += 16) {
    const int xr = rb + l15;                                     // this lane's X row as the B operand's column
    const bool x_ok = xr < rows;
    const bf16_t *xrow = X + (int64_t)(r0 + min(xr, rows - 1)) * K;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = kbeg; k0 < kend; k0 += 32 * U) {
      frag a[U], b[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = k0 + u * 32 + kc;                          // (K % 8 == 0: a chunk is inside the row or wholly past it)
        const bool ok = k < kend;
        a[u] = (ok && w_ok) ? *reinterpret_cast<const frag *>(wrow + k) : zero;
        b[u] = (ok && x_ok) ? *reinterpret_cast<const frag *>(xrow + k) : zero;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) mma(acc, a[u], b[u]);
    }
    if constexpr (KS > 1) {
      if (rb > 0) __syncthreads();                               // (the previous row block's partials have been read)
      if (ks > 0) s_part[ks - 1][lane] = acc;
      __syncthreads();
      if (ks == 0) {
#pragma unroll
        for (int w2 = 0; w2 < KS - 1; ++w2) { const f32x4 t = s_part[w2][lane]; acc[0] += t[0]; acc[1] += t[1]; acc[2] += t[2]; acc[3] += t[3]; }
      }
    }
    // D rows = output columns nq + q, D column = X row l15
    if (ks == 0 && x_ok && nq < N) {
      TO o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v = acc[q] + bq[q];
        v = to_f32(from_f32<TO>(v));
        v = act_fwd<true>(v, act);
        o[q] = from_f32<TO>(v);
      }
      TO *dst = C + (int64_t)(r0 + xr) * N + nq;
      if (nq + 3 < N) *reinterpret_cast<uint2 *>(dst) = *reinterpret_cast<const uint2 *>(o);
      else for (int q = 0; q < 4 && nq + q < N; ++q) dst[q] = o[q];
    }
  }
}

// ------------------------------------------------------------------------------------------
// 256 x 256 NT kernel, persistent, FOUR-slot ring of 32-deep stages, epilogue straight from the accumulators (round 4).
// What round 4's stamps of the two-per-CU kernel say: with the staging pass gone its K loop takes what the epilogue gave
// back (15 -> 18-19 us per 256 x 128 tile at K = 704 whatever the epilogue) - two work-groups with 48 KiB of LDS-DMA in
// flight each pull ~57 GB/s per CU from L2, which is what the chip's L2 -> LDS path gives (MI355X_MICROARCH.md: 66-73 GB/s per
// CU for gathers from L2), and a 256 x 128 tile needs 24 KiB per 32-deep sub-step: the K loop is FILL-bound at 0.0117 B/flop.
// The persistent kernels above have the better tile (0.0078 B/flop) but ONE 64-76 KiB stage in flight per CU (~35 GB/s).
// Here: one work-group of 8 waves per CU, tile 256 x 256, wave tile 64 rows x 128 columns (4 x 2 waves; X on the MFMA A
// operand, W rows permuted per 128-column half as nt2x_epilogue wants them), the two-per-CU kernel's LDS image (64-byte rows,
// chunk position c ^ F[(r >> 2) & 3]) in 32 KiB stages, FOUR slots = three stages (96 KiB) in flight behind the one being
// read, a wave's four DMA pieces of a stage going out one by one behind the first four MFMA groups (round 3: a wave sits in
// each buffer_load ... lds until the address unit has taken it).  The ring runs THROUGH tile boundaries: the stream of
// (tile, sub-step) stages is filled four ahead of the one being multiplied, so the next tile's first stages land under the
// current tile's last sub-steps and its epilogue; the epilogue needs no LDS and no barrier, and the in-order vmcnt is
// accounted for by hand (the epilogue's S stores are younger than the next tile's stages 0..3 and older than its stage 4).
// ------------------------------------------------------------------------------------------
constexpr int NT4 = 512, BM4 = 256, BN4 = 256, ROWB4 = 64;
constexpr int SLOT4 = (BM4 + BN4) * ROWB4;   // 32 KiB: X rows, then W rows
constexpr int RING4 = 4 * SLOT4;

struct Tile4 { int valid, e, rows_valid, n0, cols_valid; int64_t row0; };

template <typename TO, bool RAGGED = false, bool DYN = false>   // DYN: the dynamic tile queue (its own instantiation: as a run-time
// switch its bookkeeping cost the static walk 5 % - 1229 against 1165 us per call, profiles/r6_nt4r_queue_static_cost.log)
__global__ void __launch_bounds__(NT4)
grouped_gemm_nt4r_k(const bf16_t *__restrict__ X, const bf16_t *__restrict__ W, const float *__restrict__ bias,
                    const int32_t *__restrict__ offsets, TO *__restrict__ C, TO *__restrict__ pre_act,
                    const TO *__restrict__ mul_pre, int N, int K, int ldw, int E, int n_tiles, int total_tiles, int act_flags,
                    float drop_p, uint64_t seed, int walk_g, int walk_nb, int *__restrict__ queue) {
  typedef bf16x8 frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const bool save_grad = (act_flags & APERTIS_ACT_SAVE_GRAD) != 0, mul_saved = (act_flags & APERTIS_ACT_MUL_SAVED) != 0;
  const int act = act_flags & 0xff;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 15, fg = lane >> 4;
  const int nk = (K + 31) / 32;   // >= 5 (launcher): the fill pointer enters the next tile four sub-steps before this one ends
  const int ldb = K * 2, ldwb = ldw * 2;
  const int G = gridDim.x;
  // stores a wave leaves per tile (16 rows x one 16-byte piece per output): what the hand-kept vmcnt must let pass
  const int S = pre_act ? 32 : 16;   // (two outputs, or a pre-activation pass and the activation pass)

  // virtual tile index -> tile (the two-per-CU kernel's order: an XCD takes a contiguous run, n-panel-stationary inside)
  auto decode = [&](int v) -> Tile4 {
    Tile4 t; t.valid = 0; t.e = 0; t.rows_valid = 0; t.n0 = 0; t.cols_valid = 0; t.row0 = 0;
    const int tile = xcd_remap(v, total_tiles);
    int mt, ntile;
    tile_walk(tile, total_tiles / n_tiles, n_tiles, walk_g, walk_nb, mt, ntile);
    int accm = 0;
    for (int g = 0; g < E; ++g) {
      const int r0 = offsets[g], r1 = offsets[g + 1];
      const int nt = (r1 - r0 + BM4 - 1) / BM4;
      if (mt < accm + nt) {
        const int m0 = (mt - accm) * BM4;
        t.valid = 1; t.e = g; t.row0 = r0 + m0; t.rows_valid = min(BM4, r1 - r0 - m0);
        break;
      }
      accm += nt;
    }
    t.e = __builtin_amdgcn_readfirstlane(t.e);
    t.valid = __builtin_amdgcn_readfirstlane(t.valid);
    t.rows_valid = __builtin_amdgcn_readfirstlane(t.rows_valid);
    t.row0 = (int64_t)__builtin_amdgcn_readfirstlane((int)t.row0);
    t.n0 = ntile * BN4; t.cols_valid = min(BN4, N - t.n0);
    return t;
  };
  int vnext = blockIdx.x;
  auto next_valid = [&]() -> Tile4 {
    Tile4 t; t.valid = 0; t.e = 0; t.rows_valid = 0; t.n0 = 0; t.cols_valid = 0; t.row0 = 0;
    while (vnext < total_tiles) {
      t = decode(vnext);
      vnext += G;
      if (t.valid) break;
    }
    return t;
  };
  // ---- dynamic tile queue (queue != NULL: data-parallel steps, where an RCCL kernel may hold CUs - see ntq_next).  After its
  // first, static tile a work-group draws virtual tile indices from ITS XCD's counter: ticket k of XCD x is
  // v = (1 + k / 32) * G + (k % 32) * 8 + x - exactly the indices the static walk gives that XCD's work-groups, in the same order,
  // so the tiles in flight on an XCD are still neighbours of the walk (launcher: G % 8 == 0, nk >= 11).  The ticket for the tile
  // AFTER the next one would stall the ring if it were fetched with a compiler-tracked atomic (hipcc waits vmcnt(0) for its
  // result: every stage in flight); wave 0 issues it by hand at the top of a tile - one more operation in the in-order vmcnt
  // queue, younger than the epilogue's stores: wave 0's waits of sub-steps 0..2 let one more operation pass, sub-step 3's
  // vmcnt(8) retires it - hands it round through LDS behind sub-step 3, and every wave decodes it behind sub-step 5 (two
  // barriers later), well before the fill pointer enters the next tile (sub-step nk - 5).  A ticket past the XCD's share (or
  // on one of the few padding tiles at the walk's end) takes the slow path: thread 0 steals from the other XCDs' counters
  // with ordinary atomics behind a full barrier - the kernel's tail only.
  constexpr bool dyn = DYN;
  const int home = (int)(blockIdx.x & 7), q_per = G >> 3;
  int mt_valid = 0;
  if constexpr (dyn)
    for (int g = 0; g < E; ++g) mt_valid += (offsets[g + 1] - offsets[g] + BM4 - 1) / BM4;
  auto ticket_v = [&](int x, int k) -> int { return (1 + k / q_per) * G + (k % q_per) * 8 + x; };
  auto tile_ok = [&](int v) -> bool {
    if (v >= total_tiles) return false;
    int mt, ntile;
    tile_walk(xcd_remap(v, total_tiles), total_tiles / n_tiles, n_tiles, walk_g, walk_nb, mt, ntile);
    return mt < mt_valid;
  };
  int *s_tk = reinterpret_cast<int *>(smem + RING4 + 8 * 1024);   // [0] the ticket wave 0 drew, [1] the slow path's result, [2] a dump word
  auto steal = [&]() -> int {   // thread 0 only: the next valid virtual index from any XCD's counter, -1 when all are drained
    for (int i = 0; i < 8; ++i) {
      const int x = (home + i) & 7;
      while (true) {
        const int v = ticket_v(x, atomicAdd(queue + x * NTQ_STRIDE, 1));
        if (v >= total_tiles) break;          // this XCD's share is gone (tickets grow)
        if (tile_ok(v)) return v;
      }
    }
    return -1;
  };
  // EVERY wave executes the instruction, under an exec mask set inside the asm - lane 0 in wave 0, NO lane in waves 1..7 (a
  // no-op there) - so that the code has no branch around it: behind a C-level `if (wave == 0)` hipcc merged the result into
  // another register at the join right behind the atomic, i.e. copied a register whose data had not arrived yet (the
  // destination is written when the atomic RETURNS; its only reader is the hand-written ds_write of the hand-over behind
  // sub-step 3's wait; tools/check_nt4r_ticket_isa.py checks the compiled kernels for exactly that).  (A first form had waves
  // 1..7 add to dummy words beside the counter, to give all waves one vmcnt sequence: no faster, 8x the atomics.)
  int tkv;
  auto issue_ticket = [&]() {
    const uint64_t addr = (uint64_t)(uintptr_t)(queue + home * NTQ_STRIDE);
    const int mask = __builtin_amdgcn_readfirstlane(wave == 0 ? 1 : 0);
    uint64_t saved_exec;
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b32 exec_lo, %4\n\ts_mov_b32 exec_hi, 0\n\tglobal_atomic_add %0, %2, %3, off sc0\n\ts_mov_b64 exec, %1"
                 : "=v"(tkv), "=&s"(saved_exec) : "v"(addr), "v"(1), "s"(mask) : "memory");
  };

  Tile4 cur = next_valid();
  if constexpr (dyn) if (!cur.valid) {   // (the static first tile fell on padding: nothing is in flight yet, the slow path costs nothing)
    if (tid == 0) s_tk[1] = steal();
    __syncthreads();
    const int v = __builtin_amdgcn_readfirstlane(s_tk[1]);
    if (v >= 0) cur = decode(v);
    __syncthreads();
  }
  if (!cur.valid) return;

  // ---- the fill pointer: (tile, sub-step) of the next stage to issue, and that tile's descriptors.  Past the last tile the
  // descriptors are EMPTY (every lane out of range: zeros into a slot nobody reads, no memory traffic), so every sub-step of
  // the stream issues its four pieces and the in-order vmcnt bookkeeping is the same constant everywhere. ----
  const int fsw = (4 - ((lane >> 4) & 3)) & 3;                       // F[(row >> 2) & 3] for row = lane >> 2
  const uint32_t vx0 = (uint32_t)((wave * 32 + (lane >> 2)) * ldb + (((lane & 3) ^ fsw) << 4));
  // W piece p (16 LDS rows p*16 + c, c = lane >> 2): half = p >> 3, j = p & 7 hold W rows half*128 + c*8 + j
  const uint32_t vw0 = (uint32_t)((lane >> 2) * 8 * ldwb + (((lane & 3) ^ fsw) << 4));
  const uint32_t wp0 = (uint32_t)((((wave * 2) >> 3) * 128 + ((wave * 2) & 7)) * ldwb);       // this wave's W pieces 2w, 2w+1
  const uint32_t lds0 = lds_addr_of(smem);
  v4i fxrs, fwrs, fbrs;
  int fs = 0, fvalid = 1;
  auto set_fill = [&](const Tile4 &t) {
    fxrs = raw_buffer_rsrc(X + t.row0 * K, t.valid ? (uint32_t)t.rows_valid * (uint32_t)ldb : 0u);
    fwrs = raw_buffer_rsrc(W + ((int64_t)t.e * N + t.n0) * ldw, t.valid ? (uint32_t)t.cols_valid * (uint32_t)ldwb : 0u);
    if (bias) fbrs = raw_buffer_rsrc(bias + (int64_t)t.e * N + t.n0, t.valid ? (uint32_t)t.cols_valid * 4u : 0u);
    fvalid = t.valid;
  };
  set_fill(cur);
  // The tile's bias segment (256 floats, zeros past the last column) comes in as ONE more piece in front of the tile's stage 0,
  // into a 1 KiB area of the issuing wave's own (no barrier: a wave reads only what it fetched): a compiler-tracked global load
  // in the epilogue would be waited for with vmcnt(0) - the compiler does not count the hand-issued pieces - i.e. until the
  // next tile's stages have all landed.  It is older than stage 0's pieces (landed when they are) and makes the hand-kept
  // counts one short where stage 0 of the next tile is among the younger ones: the waits there are one operation stronger.
  const uint32_t bias_lds = lds0 + RING4 + wave * 1024;
  auto issue_bias = [&]() { lds_dma16(fbrs, bias_lds, (uint32_t)lane * 16u); };
  // piece q of this wave's share of the fill stage: q = 0,1 its X pieces (rows wave*32 + q*16 ..), q = 2,3 its W pieces
  auto issue_piece = [&](uint32_t slot_off, int q) {
    const uint32_t kb = (uint32_t)fs * ROWB4, base = lds0 + slot_off;
    const uint32_t kv = RAGGED ? kb : 0u, ks = RAGGED ? 0u : kb;
    if (q < 2) lds_dma16s(fxrs, base + (wave * 2 + q) * 1024, vx0 + (uint32_t)(q * 16 * ldb) + kv, ks);
    else lds_dma16s(fwrs, base + BM4 * ROWB4 + (wave * 2 + (q - 2)) * 1024, vw0 + wp0 + (uint32_t)((q - 2) * ldwb) + kv, ks);
  };
  Tile4 nxt;   // the tile after cur (decoded at the top of cur, entered by the fill pointer four sub-steps before cur ends)
  nxt.valid = 0; nxt.e = 0; nxt.rows_valid = 0; nxt.n0 = 0; nxt.cols_valid = 0; nxt.row0 = 0;

  f32x4 acc[4][8];
  const int frd = frow * ROWB4 + ((fg ^ ((4 - ((frow >> 2) & 3)) & 3)) << 4);
  const char *abase = smem + wm * 64 * ROWB4 + frd, *bbase = smem + (BM4 + wn * 128) * ROWB4 + frd;
  frag af[2][4], bfr[8];

  // 32 MFMAs on (acur, bfr) while the fragments of the stream's next sub-step are read under them (W in place once its four
  // MFMAs have issued, X into the other set - past a tile's last sub-step they are the next tile's first) and the four pieces
  // of the fill stage go out behind the first four MFMA groups (round 3: a wave sits in each `buffer_load ... lds` until the
  // address unit has taken it).  Straight-line on purpose: a uniform branch inside the run joins control flow, and hipcc then
  // waits for the LDS reads in flight at every join (lgkmcnt(0) in front of the next MFMA group).
  // LATE: the pieces behind the LAST four groups - waves 4-7, so that the two waves of a SIMD (w and w + 4, which run this
  // code in lock-step behind the barrier) are not both parked in the address unit's queue while the matrix pipe idles.
  auto sub_step = [&](const frag (&acur)[4], frag (&anxt)[4], int nxt_off, uint32_t fill_off, bool late) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j == 0) { anxt[0] = *reinterpret_cast<const frag *>(abase + nxt_off); anxt[1] = *reinterpret_cast<const frag *>(abase + nxt_off + 16 * ROWB4); }
      if (j == 1) { anxt[2] = *reinterpret_cast<const frag *>(abase + nxt_off + 32 * ROWB4); anxt[3] = *reinterpret_cast<const frag *>(abase + nxt_off + 48 * ROWB4); }
#pragma unroll
      for (int i = 0; i < 4; ++i) mma(acc[i][j], acur[i], bfr[j]);
      bfr[j] = *reinterpret_cast<const frag *>(bbase + nxt_off + j * 16 * ROWB4);
      __builtin_amdgcn_sched_barrier(0);
      // (a wave-uniform branch around the hand-issued piece only: nothing the compiler tracks is pending differently on its
      // two sides, so the join costs no wait)
      if (late == (j >= 4)) issue_piece(fill_off, j & 3);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // (measured at B = 44, K = 704: 1037-1045 us with the pieces of waves 4-7 behind the LAST four groups, 1013 without; wave
  //  priorities move time between the K loop and the epilogue, not the sum - profiles/r4_probe_nt4r_stagger.log)
#define NT4R_STAGGER 0
#define NT4R_BARRIER lds_barrier();
#define NT4R_PRIO 0    // 0: every wave at priority 1 inside the MFMA run; 1: waves 4-7 at priority 1 throughout, waves 0-3 at 0
  // (a per-wave branch between an EARLY and a LATE copy of the whole run made hipcc spill the accumulators: 528 B of scratch)
  const bool late = NT4R_STAGGER && wave >= 4;
  if (NT4R_PRIO == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);

  // prologue: the first four stages (nk >= 5: all of cur), stage 0's fragments
  if (bias) issue_bias();
#pragma unroll 1
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_piece((uint32_t)(k * SLOT4), q);
    ++fs;   // (< nk)
  }
  wait_vmcnt<12>();
  lds_barrier();
  int cur_off = 0;      // LDS offset of the slot of the sub-step about to be multiplied
  int prev_stores = 0;  // S once an epilogue has run

  const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const uint32_t thresh16 = (uint32_t)(drop_p * 65536.f);
  for (;;) {
    if constexpr (dyn) { nxt.valid = 0; issue_ticket(); }   // (handed over behind sub-step 3, decoded behind sub-step 5)
    else nxt = next_valid();
    // Stage 0's fragments.  (A tile's last sub-step has read them already, like every sub-step reads its successor's, but
    // 48 registers held across the epilogue leave its arithmetic no room to interleave: they are read again here - the slot is
    // not refilled before sub-step 0's barrier - and the stage has landed: the wait + barrier of the prologue resp. of the
    // previous tile's last sub-step.)
#pragma unroll
    for (int i = 0; i < 4; ++i) af[0][i] = *reinterpret_cast<const frag *>(abase + cur_off + i * 16 * ROWB4);
#pragma unroll
    for (int j = 0; j < 8; ++j) bfr[j] = *reinterpret_cast<const frag *>(bbase + cur_off + j * 16 * ROWB4);
    // this tile's bias: the piece in front of its stage 0 has landed with it
    float bv[8];
    {
      const float4 b0 = bias ? *reinterpret_cast<const float4 *>(smem + RING4 + wave * 1024 + (wn * 128 + frow * 8) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 b1 = bias ? *reinterpret_cast<const float4 *>(smem + RING4 + wave * 1024 + (wn * 128 + frow * 8) * 4 + 16) : make_float4(0.f, 0.f, 0.f, 0.f);
      bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // top of sub-step s: this wave holds the fragments of s; stage s+1 of the stream must have landed - two stages are younger
    // (8 pieces), and through the first three sub-steps behind an epilogue so are its S stores; behind the barrier stage s+1 is
    // complete everywhere and every wave has read s, so the fill stage (s + 4 of the stream) may overwrite slot(s)
#define SUB4(S_, AC, AN)                                                                                       \
    {                                                                                                          \
      const int nxt_off = cur_off + SLOT4 == RING4 ? 0 : cur_off + SLOT4;                                      \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
      const int stores_behind = (S_) < 3 ? prev_stores : 0;                                                    \
      if ((S_) < 3 && dyn && wave == 0) {   /* wave 0's ticket atomic behind the stores: one more operation may pass there */ \
        if (stores_behind == 0) wait_vmcnt<9>(); else if (stores_behind == 16) wait_vmcnt<25>(); else wait_vmcnt<41>(); \
      } else if (stores_behind == 0) wait_vmcnt<8>(); else if (stores_behind == 16) wait_vmcnt<24>(); else wait_vmcnt<40>(); \
      NT4R_BARRIER                                                                                             \
      if (fs == 0 && bias && fvalid) issue_bias();                                                             \
      if (NT4R_PRIO == 0) __builtin_amdgcn_s_setprio(1);                                                       \
      sub_step(AC, AN, nxt_off, (uint32_t)cur_off, late);                                                      \
      if (NT4R_PRIO == 0) __builtin_amdgcn_s_setprio(0);                                                       \
      if (++fs == nk) { fs = 0; set_fill(nxt); }                                                               \
      cur_off = nxt_off;                                                                                       \
    }
    for (int s = 0; s < nk; s += 2) {
      SUB4(s, af[0], af[1])
      if (s + 1 < nk) SUB4(s + 1, af[1], af[0])
      else {   // odd nk: keep the register roles of the loop
#pragma unroll
        for (int i = 0; i < 4; ++i) af[0][i] = af[1][i];
      }
      if constexpr (dyn) {
        if (s == 2) {          // behind sub-step 3: its vmcnt(8) has retired wave 0's ticket atomic
          // (wave 0 / lane 0's value is the ticket; every other lane writes what its register holds to a word nobody reads)
          asm volatile("ds_write_b32 %0, %1" :: "v"(lds0 + (uint32_t)(RING4 + 8 * 1024) + ((wave | lane) ? 8u : 0u)), "v"(tkv) : "memory");
        } else if (s == 4) {   // behind sub-step 5: two barriers after the hand-over (nk >= 11: the fill pointer is still in cur)
          const int v = ticket_v(home, __builtin_amdgcn_readfirstlane(s_tk[0]));
          if (tile_ok(v)) nxt = decode(v);
          else {               // past this XCD's share, or padding: the slow path (a full barrier: every stage drains - tail only)
            __syncthreads();
            if (tid == 0) s_tk[1] = steal();
            __syncthreads();
            const int v2 = __builtin_amdgcn_readfirstlane(s_tk[1]);
            if (v2 >= 0) nxt = decode(v2);
          }
        }
      }
    }
#undef SUB4
    {
      const int n0w = cur.n0 + wn * 128, cvw = cur.cols_valid - wn * 128;
#define OUT(MODE, A, D, DST, DST2) \
  nt2x_epilogue<TO, MODE, A, D>(acc, bv, DST, DST2, mul_pre, cur.row0, cur.rows_valid, n0w, cvw, N, act, drop_p, seed, keep_scale, thresh16, wm, frow, fg)
      if (save_grad && pre_act) {   // (GELU: launch_nt refuses the flag for other activations)
        if (drop_p > 0.f) OUT(EPI_BOTH, APERTIS_ACT_GELU, true, C, pre_act);
        else OUT(EPI_BOTH, APERTIS_ACT_GELU, false, C, pre_act);
      } else {
        if (pre_act) OUT(EPI_RAW, APERTIS_ACT_NONE, false, pre_act, nullptr);
        if (mul_pre) {   // (no bias beside mul_pre: launch_nt)
#define OUTNB(MODE, A, D, DST, DST2) \
  nt2x_epilogue<TO, MODE, A, D, false>(acc, bv, DST, DST2, mul_pre, cur.row0, cur.rows_valid, n0w, cvw, N, act, drop_p, seed, keep_scale, thresh16, wm, frow, fg)
          if (mul_saved) OUTNB(EPI_MULSAVED, APERTIS_ACT_NONE, false, C, nullptr);
          else if (act == APERTIS_ACT_GELU && drop_p > 0.f) OUTNB(EPI_MULACT, APERTIS_ACT_GELU, true, C, nullptr);
          else if (act == APERTIS_ACT_GELU) OUTNB(EPI_MULACT, APERTIS_ACT_GELU, false, C, nullptr);
          else OUTNB(EPI_MULACT, -1, false, C, nullptr);
#undef OUTNB
        } else if (act == APERTIS_ACT_NONE && drop_p <= 0.f) OUT(EPI_RAW, APERTIS_ACT_NONE, false, C, nullptr);
        else if (act == APERTIS_ACT_GELU && drop_p > 0.f) OUT(EPI_ACT, APERTIS_ACT_GELU, true, C, nullptr);
        else if (act == APERTIS_ACT_GELU) OUT(EPI_ACT, APERTIS_ACT_GELU, false, C, nullptr);
        else OUT(EPI_ACT, -1, false, C, nullptr);
      }
#undef OUT
    }
    if (!nxt.valid) break;
    cur = nxt;
    prev_stores = S;
  }
  // (the empty pieces of the stream's last sub-steps are still in flight: they write zeros into this work-group's own LDS)
  wait_vmcnt<0>();
}

// dpre = dh * keepmask/(1-p) * act'(pre)   (elementwise, rows < offsets[E])
template <typename T>
__global__ void act_dropout_bwd_k(const T *__restrict__ dh, const T *__restrict__ pre, T *__restrict__ dpre,
                                  const int32_t *__restrict__ offsets, int64_t max_rows, int N, int E, int act,
                                  float drop_p, uint64_t seed) {
  const int64_t total_rows = min((int64_t)offsets[E], max_rows);
  const int64_t nvec = total_rows * N / 4;
  const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  const uint32_t thresh16 = (uint32_t)(drop_p * 65536.f);
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
    const int64_t lin = v * 4;
    const int64_t row = lin / N;
    const int col = (int)(lin - row * N);
    float d[4], p[4], o[4];
    if constexpr (sizeof(T) == 4) {
      float4 a = *reinterpret_cast<const float4 *>(dh + lin), b = *reinterpret_cast<const float4 *>(pre + lin);
      d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; p[0] = b.x; p[1] = b.y; p[2] = b.z; p[3] = b.w;
    } else {
      typedef __attribute__((ext_vector_type(4))) bf16_t bf4;
      bf4 a = *reinterpret_cast<const bf4 *>(dh + lin), b = *reinterpret_cast<const bf4 *>(pre + lin);
#pragma unroll
      for (int j = 0; j < 4; ++j) { d[j] = (float)a[j]; p[j] = (float)b[j]; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float g = d[j] * act_grad<sizeof(T) == 2>(p[j], act);
      if (drop_p > 0.f) g = gd_keep(seed, row, col + j, thresh16) ? g * keep_scale : 0.f;
      o[j] = g;
    }
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<float4 *>(dpre + lin) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
      typedef __attribute__((ext_vector_type(4))) bf16_t bf4;
      bf4 r = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
      *reinterpret_cast<bf4 *>(dpre + lin) = r;
    }
  }
}

// ------------------------------------------------------------------------------------------
// TN (weight gradient):  dW[e][m][n] = sum_{r in group e} A[r][m] * Bm[r][n]
// K dimension = the group's rows.  Both operands are K-strided in memory, so tiles are staged
// as [k][row] images (k = token row, 64 bf16 / 32 fp32 per step) and fragments are gathered
// transposed: bf16 with ds_read_b64_tr_b16, fp32 with scalar LDS reads.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(NT)
grouped_gemm_tn_k(const T *__restrict__ A, const T *__restrict__ Bm, const int32_t *__restrict__ offsets,
                  float *__restrict__ dW, float *__restrict__ dbias, int M, int N, int m_tiles, int n_tiles) {
  // LDS images: As[k][BM] and Bs[k][BN], k = 32 rows per step, element rows of 128*sizeof(T) bytes
  constexpr int BKR = 32;                       // token rows per step
  constexpr int PITCH = BM * sizeof(T) + 16;    // bytes per k-row (+16 pad against bank conflicts)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *As = smem, *Bs = smem + BKR * PITCH;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int e = tile / (m_tiles * n_tiles);
  const int rem = tile - e * m_tiles * n_tiles;
  const int mt = rem / n_tiles, ntile = rem - mt * n_tiles;
  const int m0 = mt * BM, n0 = ntile * BN;
  const int r_begin = offsets[e], r_end = offsets[e + 1];
  const int mvalid = min(BM, M - m0), nvalid = min(BN, N - n0);

  f32x4 acc[4][4];  // acc[i][j]: D rows = m-subtile i (from A), D cols = n-subtile j (from Bm)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;  // column sum of A for dbias (threads 0..BM-1 of the ntile==0 blocks)

  constexpr int EPC = 16 / sizeof(T);            // elements per 16-byte chunk
  constexpr int CPRW = BM / EPC;                 // chunks per k-row
  constexpr int LOADS = BKR * CPRW / NT;         // 16-byte loads per thread per operand
  const int frow = lane & 15, fg = lane >> 4;

  for (int r0 = r_begin; r0 < r_end; r0 += BKR) {
    uint4 ar[LOADS], br[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      int q = tid + i * NT;
      int kr = q / CPRW, c = q % CPRW;
      int col = c * EPC;
      bool rok = r0 + kr < r_end;
      ar[i] = (rok && col < mvalid) ? *reinterpret_cast<const uint4 *>(A + (int64_t)(r0 + kr) * M + m0 + col)
                                    : make_uint4(0, 0, 0, 0);
      br[i] = (rok && col < nvalid) ? *reinterpret_cast<const uint4 *>(Bm + (int64_t)(r0 + kr) * N + n0 + col)
                                    : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();  // previous step's fragment reads are done
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      int q = tid + i * NT;
      int kr = q / CPRW, c = q % CPRW;
      *reinterpret_cast<uint4 *>(As + kr * PITCH + c * 16) = ar[i];
      *reinterpret_cast<uint4 *>(Bs + kr * PITCH + c * 16) = br[i];
    }
    __syncthreads();
    if (dbias && ntile == 0 && tid < BM) {
#pragma unroll 8
      for (int kr = 0; kr < BKR; ++kr) bsum += to_f32(*reinterpret_cast<const T *>(As + kr * PITCH + tid * sizeof(T)));
    }
    if constexpr (sizeof(T) == 2) {
      // one MFMA 16x16x32 consumes all 32 k-rows: lane (frow, fg) needs k = 8*fg .. 8*fg+7 for
      // its row/col; two transposed reads of 4 k-rows x 16 columns each
      typedef __attribute__((ext_vector_type(4))) short s16x4;
      bf16x8 af[4], bf[4];
      const int lr = lane & 15;           // lane within its 16-lane group
      const int trow = lr >> 2, tcol4 = (lr & 3) * 4;  // address this lane supplies: row q, cols 4p..4p+3
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        union { bf16x8 v; s16x4 h[2]; } ua, ub;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          int krow = 8 * fg + 4 * hh + trow;
          const char *pa = As + krow * PITCH + (wm * 64 + i * 16 + tcol4) * 2;
          const char *pb = Bs + krow * PITCH + (wn * 64 + i * 16 + tcol4) * 2;
          ua.h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)pa);
          ub.h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)pb);
        }
        af[i] = ua.v; bf[i] = ub.v;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mma(acc[i][j], af[i], bf[j]);
    } else {
      // fp32: MFMA 16x16x4 step s: lane supplies A[m=frow][k=4s+fg], B[k=4s+fg][n=frow]
#pragma unroll
      for (int s = 0; s < BKR / 4; ++s) {
        const int krow = 4 * s + fg;
        float af[4], bf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          af[i] = *reinterpret_cast<const float *>(As + krow * PITCH + (wm * 64 + i * 16 + frow) * 4);
          bf[i] = *reinterpret_cast<const float *>(Bs + krow * PITCH + (wn * 64 + i * 16 + frow) * 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
  }
  // D layout: col = lane&15 -> n, row = fg*4+reg -> m
  float *out = dW + (int64_t)e * M * N;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int m = m0 + wm * 64 + i * 16 + fg * 4 + r, n = n0 + wn * 64 + j * 16 + frow;
        if (m < M && n < N) out[(int64_t)m * N + n] = acc[i][j][r];
      }
  if (dbias && ntile == 0 && tid < mvalid) dbias[(int64_t)e * M + m0 + tid] = bsum;
}

// ------------------------------------------------------------------------------------------
// TN v2 (bf16): 128 x 128 output tile per (expert, m-tile, n-tile), K = the expert's rows in
// steps of 64.  Operand images As[64 k][128 m], Bs[64 k][128 n] arrive by LDS-DMA into a
// double-buffered ring (2 x 32 KiB), full K steps only; the last partial step is staged through
// registers with zero fill.  k-rows are 256 B = one LDS bank row, so the transposed fragment
// reads (ds_read_b64_tr_b16: 4 k-rows x 32 B per 16-lane group, two groups per half-wave) would
// be 8-way conflicted; the 32-byte window index is XORed with f(k) = (k&3)|((k>>3)&1)<<2 on the
// DMA source address and on the read, which spreads the 8 rows of a half-wave over the 8 windows.
// Bm feeds the MFMA A operand, A the B operand: a lane's accumulator holds 4 consecutive n of one
// m, so dW leaves through an LDS staging tile as 16-byte row pieces.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int tn_swz(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }

// One launch can carry TWO weight-gradient problems over the same row grouping (the two expert
// layers): 2 x 1056 tiles fill 4.1 rounds of the chip instead of 2 x 3 (tile-count quantisation
// is the largest loss of this kernel at E=8, H=704, I=2816).
struct TnProblem {
  const bf16_t *A, *Bm;
  float *dW, *dbias;
  int M, N, m_tiles, n_tiles, tiles;
};

__global__ void __launch_bounds__(NT)
grouped_gemm_tn2_k(TnProblem p0, TnProblem p1, const int32_t *__restrict__ offsets) {
  typedef bf16_t T;
  constexpr int BKR = 64;              // rows (K) per step
  constexpr int KROWB = BM * 2;        // 256 B per k-row
  constexpr int STAGE = 2 * BKR * KROWB;  // As + Bs = 32 KiB
  constexpr int CP = BN * 4 + 16;      // fp32 C staging pitch
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((ext_vector_type(4))) short s16x4;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int tile = xcd_remap(blockIdx.x, gridDim.x);
  const bool second = tile >= p0.tiles;
  if (second) tile -= p0.tiles;
  const TnProblem &pp = second ? p1 : p0;
  const bf16_t *__restrict__ A = pp.A, *__restrict__ Bm = pp.Bm;
  float *__restrict__ dW = pp.dW, *__restrict__ dbias = pp.dbias;
  const int M = pp.M, N = pp.N, m_tiles = pp.m_tiles, n_tiles = pp.n_tiles;
  const int e = tile / (m_tiles * n_tiles);
  const int rem = tile - e * m_tiles * n_tiles;
  // the SHORTER tile dimension runs fastest: the ~64 tiles an XCD runs at once then form a block
  // about as wide as it is tall and share the fewest distinct operand column blocks in its L2
  // (PMC: 3.3x over-fetch with the long dimension fastest)
  int mt, ntile;
  if (m_tiles < n_tiles) { ntile = rem / m_tiles; mt = rem - ntile * m_tiles; }
  else { mt = rem / n_tiles; ntile = rem - mt * n_tiles; }
  const int m0 = mt * BM, n0 = ntile * BN;
  const int r_begin = offsets[e], r_end = offsets[e + 1];
  const int mvalid = min(BM, M - m0), nvalid = min(BN, N - n0);
  const int nfull = (r_end - r_begin) / BKR;
  const int tail = (r_end - r_begin) - nfull * BKR;
  const int nsteps = nfull + (tail ? 1 : 0);

  f32x4 acc[4][4];  // [n-subtile j][m-subtile i]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;

  // LDS-DMA of one full K step: 16 pieces per operand (4 k-rows each), 4 per wave
  auto stage = [&](int buf, int step) {
    char *as = smem + buf * STAGE, *bs = as + BKR * KROWB;
    const int64_t r0 = (int64_t)r_begin + (int64_t)step * BKR;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int piece = wave * 4 + j;
      const int krow = piece * 4 + (lane >> 4);
      const int cpos = lane & 15;
      const int csrc = (((cpos >> 1) ^ tn_swz(krow)) << 1) | (cpos & 1);
      const int ca = min(csrc * 8, mvalid - 8), cb = min(csrc * 8, nvalid - 8);   // clamp column chunk (masked at store)
      const T *pa = A + (r0 + krow) * M + m0 + ca;
      const T *pb = Bm + (r0 + krow) * N + n0 + cb;
      lds_dma16_global(pa, lds_addr_of(as + piece * 1024));
      lds_dma16_global(pb, lds_addr_of(bs + piece * 1024));
    }
  };
  // register-staged, zero-filled tail step (rows >= r_end contribute nothing)
  auto stage_tail = [&](int buf, int step) {
    char *as = smem + buf * STAGE, *bs = as + BKR * KROWB;
    const int64_t r0 = (int64_t)r_begin + (int64_t)step * BKR;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int q = tid + it * NT;            // 1024 chunks of 16 B per operand
      const int krow = q >> 4, cpos = q & 15;
      const int csrc = (((cpos >> 1) ^ tn_swz(krow)) << 1) | (cpos & 1);
      const bool rok = krow < tail;
      uint4 va = make_uint4(0, 0, 0, 0), vb = make_uint4(0, 0, 0, 0);
      if (rok && csrc * 8 < mvalid) va = *reinterpret_cast<const uint4 *>(A + (r0 + krow) * M + m0 + csrc * 8);
      if (rok && csrc * 8 < nvalid) vb = *reinterpret_cast<const uint4 *>(Bm + (r0 + krow) * N + n0 + csrc * 8);
      *reinterpret_cast<uint4 *>(as + q * 16) = va;
      *reinterpret_cast<uint4 *>(bs + q * 16) = vb;
    }
  };

  const int frow = lane & 15, fg = lane >> 4;
  const int lr = lane & 15, trow = lr >> 2, tcol4 = (lr & 3) * 4;
  // tn_swz(kb*32 + 8*fg + 4*hh + trow) = trow | ((fg & 1) << 2) for every kb, hh
  const int lane_sw = trow | ((fg & 1) << 2);
  int aoff[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    aoff[i] = (8 * fg + trow) * KROWB + ((((wm * 4 + i) ^ lane_sw) << 5) | (tcol4 * 2));
    boff[i] = (8 * fg + trow) * KROWB + ((((wn * 4 + i) ^ lane_sw) << 5) | (tcol4 * 2));
  }
  if (nsteps > 0) {
    if (nfull > 0) stage(0, 0); else stage_tail(0, 0);
  }
  for (int st = 0; st < nsteps; ++st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (st + 1 < nsteps) {
      if (st + 1 < nfull) stage((st + 1) & 1, st + 1); else stage_tail((st + 1) & 1, st + 1);
    }
    const char *as = smem + (st & 1) * STAGE, *bs = as + BKR * KROWB;
    if (dbias && ntile == 0 && tid < BM) {
      const int c = tid >> 3, within = (tid & 7) * 2;
#pragma unroll 8
      for (int kr = 0; kr < BKR; ++kr) {
        const int cp = (((c >> 1) ^ tn_swz(kr)) << 1) | (c & 1);
        bsum += to_f32(*reinterpret_cast<const T *>(as + kr * KROWB + cp * 16 + within));
      }
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {       // two 32-deep MFMA k-blocks per step
      bf16x8 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        union { bf16x8 v; s16x4 h[2]; } ua, ub;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          // k-row = kb*32 + 8*fg + 4*hh + trow: its swizzle key only depends on the lane (lane_sw), so
          // the per-lane byte offsets are loop invariants and (kb, hh) add compile-time constants
          const int kconst = (kb * 32 + 4 * hh) * KROWB;
          ua.h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4 *)(as + aoff[i] + kconst));
          ub.h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4 *)(bs + boff[i] + kconst));
        }
        af[i] = ua.v; bf[i] = ub.v;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) mma(acc[j][i], bf[j], af[i]);
    }
  }
  // epilogue: D row = n (fg*4+r within subtile j), D col = m (frow within subtile i)
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = wm * 64 + i * 16 + frow, n = wn * 64 + j * 16 + fg * 4;
      *reinterpret_cast<f32x4 *>(smem + m * CP + n * 4) = acc[j][i];
    }
  __syncthreads();
  float *out = dW + (int64_t)e * M * N;
  for (int q = tid; q < BM * (BN / 4); q += NT) {
    const int row = q >> 5, c = q & 31;
    if (row < mvalid && c * 4 < nvalid)
      *reinterpret_cast<uint4 *>(out + (int64_t)(m0 + row) * N + n0 + c * 4) =
          *reinterpret_cast<const uint4 *>(smem + row * CP + c * 16);
  }
  if (dbias && ntile == 0 && tid < mvalid) dbias[(int64_t)e * M + m0 + tid] = bsum;
}

int launch_tn2(const TnProblem &q0, const TnProblem &q1, const int32_t *offsets, hipStream_t st) {
  size_t lds = std::max<size_t>(2 * 2 * 64 * 256, (size_t)BM * (BN * 4 + 16));
  hipFuncSetAttribute((const void *)grouped_gemm_tn2_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(grouped_gemm_tn2_k, dim3((unsigned)(q0.tiles + q1.tiles)), dim3(NT), lds, st, q0, q1, offsets);
  return apertis_check_launch();
}

// ------------------------------------------------------------------------------------------
// TN v3 (bf16): 256 x 256 output tiles, 512 threads (2(M) x 4(N) waves of 128 x 64), one persistent
// work-group per CU.  What bounded v2 on the expert weight gradients (E=8, 2816 x 704): (a) tile
// count quantisation - 2 x 1056 tiles of 128^2 on 512 slots is 4.1 rounds, run as 5; (b) operand
// re-fetch - a 128^2 tile needs 64 flop per L2 byte and the 132 tiles of one (problem, expert) are
// spread over two or three XCDs, each fetching the same rows into its own L2 (PMC: 2.4x the
// algorithmic HBM bytes).  Here the CUs are dealt out in equal shares to the (problem, expert)
// groups - a group's CUs are neighbours on one XCD (xcd_remap) - and a group's tiles are processed
// in lock-step rounds of `cpg` tiles, all walking the group's rows together, so a row block is
// fetched into that XCD's L2 once per round.  The tiles left over after the full rounds are not
// given a round of their own: each is split along the rows into cpg/rem slices, one per CU, whose
// partial tiles go to the caller's workspace and are summed in slice order by tn3_fold_k
// (deterministic; no atomics).  With E=1 this is an ordinary split-K GEMM.
// Operand images As[64 k][256 m], Bs[64 k][256 n] (512-byte k-rows) arrive by buffer_load...lds
// through descriptors sized to the group's rows - rows past the end read as zero, so there is no
// tail path - double-buffered (2 x 64 KiB); the transposed fragment reads and the 32-byte window
// swizzle are v2's.  The bias gradient (column sums of A) comes from the matrix pipe as well: one
// extra MFMA per k-block against an all-ones fragment, two m-subtiles per wave.
// ------------------------------------------------------------------------------------------
struct Tn3Problem {
  const bf16_t *A, *Bm;
  float *dW, *dbias;
  int M, N, m_tiles, n_tiles;
};
struct Tn3Args {
  Tn3Problem p0, p1;
  int nprob, E, cpg;   // cpg = CUs (work-groups) per (problem, expert) group
  float *ws;           // [grid][TN3_SLOT] partial tiles (+ 256 partial bias sums each)
  int *ctr;            // [groups] item counters of the launch (zeroed in front of it), behind the partial tiles in `ws`
};
constexpr int TN3_SLOT = 256 * 256 + 256;
constexpr int TN3_CTR_STRIDE = 64;                      // ints between two groups' counters: a 256-byte line each - on one line the
                                                        // device-scope atomics of all XCDs queue up behind each other
constexpr int TN3_CTR_BYTES = 1024 * TN3_CTR_STRIDE * 4;   // item counters of the groups (groups <= #CUs <= 1024), behind the partial tiles

// the part of the schedule the GEMM and the fold kernel must agree on
struct Tn3Sched { int T, full, rem, s; };
// tile index -> (m-tile, n-tile), the dimension with FEWER tiles fastest: the cpg tiles of a round
// then cover all blocks of the narrow operand but only ~cpg/min(m_tiles, n_tiles) blocks of the wide
// one, so per round the wide operand is fetched once and only the narrow one again
__host__ __device__ inline void tn3_tile_coord(int tile, int m_tiles, int n_tiles, int &mt, int &nt) {
  if (m_tiles < n_tiles) { nt = tile / m_tiles; mt = tile - nt * m_tiles; }
  else { mt = tile / n_tiles; nt = tile - mt * n_tiles; }
}
__host__ __device__ inline Tn3Sched tn3_sched(int m_tiles, int n_tiles, int cpg) {
  Tn3Sched c;
  c.T = m_tiles * n_tiles;
  c.full = c.T / cpg;
  c.rem = c.T - c.full * cpg;
  c.s = c.rem ? cpg / c.rem : 0;
  return c;
}

template <bool RING, bool STAGGER = false>
__global__ void __launch_bounds__(NT2)
grouped_gemm_tn3_k(Tn3Args a, const int32_t *__restrict__ offsets) {
  typedef bf16_t T;
  constexpr int BKR = 64, KROWB = 512, OPB = BKR * KROWB;   // 32 KiB per operand image
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((ext_vector_type(4))) short s16x4;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int g = id / a.cpg, j = id - g * a.cpg;
  const bool second = g >= a.E;
  const int e = second ? g - a.E : g;
  const T *__restrict__ A = second ? a.p1.A : a.p0.A, *__restrict__ Bm = second ? a.p1.Bm : a.p0.Bm;
  float *__restrict__ dW = second ? a.p1.dW : a.p0.dW, *__restrict__ dbias = second ? a.p1.dbias : a.p0.dbias;
  const int M = second ? a.p1.M : a.p0.M, N = second ? a.p1.N : a.p0.N;
  const int m_tiles = second ? a.p1.m_tiles : a.p0.m_tiles, n_tiles = second ? a.p1.n_tiles : a.p0.n_tiles;
  const Tn3Sched sc = tn3_sched(m_tiles, n_tiles, a.cpg);
  const int r_begin = offsets[e], rows = offsets[e + 1] - r_begin;
  const int nsteps = (rows + BKR - 1) / BKR;
  const int ldA = M * (int)sizeof(T), ldB = N * (int)sizeof(T);

  // per-lane parts of the addresses.  DMA piece p (two k-rows) is handled by wave p & 7: the
  // swizzle key of its rows, f(k) = (k&3) | ((k>>3)&1)<<2 with k = 2p + (lane>>5), then only
  // depends on the wave and the lane
  const int hi = lane >> 5;
  const int chunk = (lane & 31) ^ (hi << 1) ^ ((wave & 1) << 2) ^ (((wave >> 2) & 1) << 3);
  const uint32_t va0 = (uint32_t)(hi * ldA + chunk * 16), vb0 = (uint32_t)(hi * ldB + chunk * 16);
  const int frow = lane & 15, fg = lane >> 4;
  const int trow = frow >> 2, tcol4 = (frow & 3) * 4;
  const int lane_sw = trow | ((fg & 1) << 2);
  const int rd0 = (8 * fg + trow) * KROWB + tcol4 * 2;
  bf16x8 ones;
#pragma unroll
  for (int q = 0; q < 8; ++q) ones[q] = (bf16_t)1.0f;
  const int wn_u = __builtin_amdgcn_readfirstlane(wn);
  const bool lagger = __builtin_amdgcn_readfirstlane(wave) >= 4;

  // A group's items in order: the full tiles round by round, then the row slices of the remainder tiles.
  //  static walk (a.ctr == NULL): work-group j takes items j, j + cpg, ... and the slice j.
  //  item queue (a.ctr): after its first item a work-group takes the next from the group's counter - one whose CU was held by
  //  another kernel when the grid started (an RCCL collective on the communication stream) then finds the group's items
  //  taken instead of walking a full static share alone.  tools/probes/hog_probe.hip, 32 of 256 CUs held for the whole
  //  launch: static 2400 us against 1460 alone, queue 1900 us; alone the queue costs 3 % (1510 us), which is why it is the
  //  caller's choice.  Which work-group computes an item changes neither its arithmetic nor where it lands: bit-identical.
  int *s_next = reinterpret_cast<int *>(smem + 4 * OPB);
  const int n_full = sc.full * a.cpg, n_items = n_full + (sc.rem ? sc.rem * sc.s : 0);
  int item = j;   // first item: the work-group's own index ...
  if (a.ctr) {
    // ... unless the items come from the group's queue: then the FIRST one does too.  A work-group whose CU another kernel
    // holds (an RCCL collective) starts when some other work-group has exited, i.e. when the queue is empty - with a
    // statically assigned first item it then ran a whole tile alone behind everyone else (tools/probes/hog_probe.hip, 32 of
    // 256 CUs held: 1900 us against 1670 for 224 CUs' worth of work; now it finds nothing and leaves)
    if (tid == 0) s_next[0] = atomicAdd(a.ctr + g * TN3_CTR_STRIDE, 1);
    __syncthreads();
    item = __builtin_amdgcn_readfirstlane(s_next[0]);   // (slot 0 is next written two barriers from here)
  }
  for (int par = 1; item < n_items; par ^= 1) {
    int nxt = 0;
    if (a.ctr && tid == 0) nxt = atomicAdd(a.ctr + g * TN3_CTR_STRIDE, 1);   // returns under the K loop
    int tile, s0 = 0, s1 = nsteps, slot = 0;
    bool partial = false;
    if (item < n_full) {
      tile = item;
    } else {
      const int r = item - n_full, ri = r / sc.s;
      const int sl = r - ri * sc.s, per = (nsteps + sc.s - 1) / sc.s;
      tile = n_full + ri;
      s0 = min(sl * per, nsteps);
      s1 = min(s0 + per, nsteps);
      partial = sc.s > 1;
      slot = g * a.cpg + r;
    }
    int mt, nt;
    tn3_tile_coord(tile, m_tiles, n_tiles, mt, nt);
    const int m0 = mt * 256, n0 = nt * 256;
    const bool want_bias = dbias != nullptr && nt == 0;
    // descriptors from the tile's first column of the group's first row to the end of its last row
    const v4i ars = raw_buffer_rsrc(A + (int64_t)r_begin * M + m0, rows > 0 ? (uint32_t)(rows * ldA - m0 * (int)sizeof(T)) : 0u);
    const v4i brs = raw_buffer_rsrc(Bm + (int64_t)r_begin * N + n0, rows > 0 ? (uint32_t)(rows * ldB - n0 * (int)sizeof(T)) : 0u);
    const uint32_t lds0 = lds_addr_of(smem);
    auto stage = [&](int buf, int step) {
      const uint32_t as = lds0 + buf * 2 * OPB, bs = as + OPB;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int p = jj * 8 + wave;
        const uint32_t row = (uint32_t)(step * BKR + 2 * p);   // in the VGPR offset: the range check ignores the scalar one
        lds_dma16(ars, as + p * 1024, va0 + row * (uint32_t)ldA);
        lds_dma16(brs, bs + p * 1024, vb0 + row * (uint32_t)ldB);
      }
    };

    f32x4 acc[4][8];   // [n-subtile][m-subtile]
#pragma unroll
    for (int jn = 0; jn < 4; ++jn)
#pragma unroll
      for (int im = 0; im < 8; ++im) acc[jn][im] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 accb[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};

    // one 32-deep k-block: transposed fragment reads from the images at (as, bs), then 32 MFMAs (+ the bias MFMAs)
    bf16x8 af[8], bf[4];
    auto read_frags = [&](const char *as, const char *bs) {
#pragma unroll
      for (int im = 0; im < 8; ++im) {
        union { bf16x8 v; s16x4 h[2]; } u;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
          u.h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(
              as + rd0 + (((wm * 8 + im) ^ lane_sw) << 5) + (4 * hh) * KROWB));
        af[im] = u.v;
      }
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) {
        union { bf16x8 v; s16x4 h[2]; } u;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
          u.h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(
              bs + rd0 + (((wn * 4 + jn) ^ lane_sw) << 5) + (4 * hh) * KROWB));
        bf[jn] = u.v;
      }
    };
    auto mma_frags = [&](auto &&between) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) {
#pragma unroll
        for (int im = 0; im < 8; ++im) mma(acc[jn][im], bf[jn], af[im]);
        between(jn);
      }
      if (want_bias) {
        // this wave's two m-subtiles, picked without indexing the register array.  wn_u is in an
        // SGPR on purpose: MFMA ignores EXEC, and hipcc drops the execz skip around a short
        // divergent block, so an `if (lane-derived wn == ...)` here runs the MFMA in every wave
#pragma unroll
        for (int im = 0; im < 8; ++im)
          if ((im >> 1) == wn_u) mma(accb[im & 1], ones, af[im]);
      }
      __builtin_amdgcn_s_setprio(0);
    };
    auto nothing = [](int) {};
    auto kblock = [&](const char *as, const char *bs) { read_frags(as, bs); mma_frags(nothing); };
    if constexpr (RING) {
      // four slots of 32 k-rows (16 KiB per operand), THREE stages in flight (96 KiB per CU instead of the 64 of the
      // double buffer below): a stage of this kernel comes from HBM - every k-row of the group is read once per round -
      // and with one 64-deep stage in flight the step took as long as the stage's latency (2.2 us against 0.9 us of
      // MFMA work).  The DMA pieces are whole 512-byte k-rows either way, so the shallower stage costs the fill nothing
      // (unlike the NT kernels, whose rows would shrink to 64 bytes).
      constexpr int OPH = 32 * KROWB;
      const int nsub = 2 * (s1 - s0);
      // piece q (0..3) of this wave's share of stage u: q & 1 picks the operand, q >> 1 the piece (two k-rows of 512 B).
      // Past the item's last stage the piece is still issued, from beyond the descriptor's range (it writes zeros into a
      // slot nobody reads again and moves no data): the loop then has ONE shape and one vmcnt count
      auto piece32 = [&](int u, int q) {
        const uint32_t as = lds0 + (u & 3) * 2 * OPH, bs = as + OPH;
        const int p = (q >> 1) * 8 + wave;
        const uint32_t row = (uint32_t)(s0 * BKR + u * 32 + 2 * p);
        const bool live = u < nsub;
        if (q & 1) lds_dma16(brs, bs + p * 1024, live ? vb0 + row * (uint32_t)ldB : 0xfffffff0u);
        else lds_dma16(ars, as + p * 1024, live ? va0 + row * (uint32_t)ldA : 0xfffffff0u);
      };
      auto stage32 = [&](int u) {
#pragma unroll
        for (int q = 0; q < 4; ++q) piece32(u, q);
      };
      for (int u = 0; u < 3; ++u) stage32(u);
      // top of interval u: stage u has landed once only the DMAs of the two younger stages are outstanding (vmcnt retires
      // in order); behind the barrier every wave's share of stage u is in and nobody still reads slot (u - 1) & 3, which
      // stage u + 3 then overwrites.  The four DMA instructions of that stage are NOT issued here in one run: a wave sits
      // in each of them until the CU's address unit has taken it (16 cycles per KiB piece, 32 pieces per interval and CU,
      // all eight waves at once right behind the barrier - the 0.4-0.5 us per 64-deep step that "asynchronous" fills cost
      // every persistent kernel of this file in round 2's probes); they go out one by one between the MFMA groups
      auto top = [&]() {
        wait_vmcnt<8>();
        lds_barrier();
      };
      auto mma_and_fill = [&](int u) {   // the MFMAs of the fragments in registers, stage u + 3 on its way between them
        mma_frags([&](int jn) {
          __builtin_amdgcn_sched_barrier(0);
          piece32(u + 3, jn);
          __builtin_amdgcn_sched_barrier(0);
        });
      };
      if (!STAGGER || !lagger) {
        for (int u = 0; u < nsub; ++u) {
          top();
          const char *as = smem + (u & 3) * 2 * OPH;
          read_frags(as, as + OPH);
          mma_and_fill(u);
        }
      } else if (nsub > 0) {
        // waves 4-7 (the SIMD partners of waves 0-3) run half an interval out of phase: the MFMAs of the fragments they
        // read in the previous interval first - while their partners, who start with the reads, leave the matrix pipe
        // alone - then this interval's reads under the partners' MFMAs (MI355X_MICROARCH.md, two waves per SIMD, item 9).
        // A loop of its own: with the role test inside one loop hipcc kept both roles' registers live (772 B of scratch)
        top();
        stage32(3);
        read_frags(smem, smem + OPH);
        for (int u = 1; u < nsub; ++u) {
          top();
          mma_and_fill(u);
          const char *as = smem + (u & 3) * 2 * OPH;
          read_frags(as, as + OPH);
        }
        mma_frags(nothing);
      }
      wait_vmcnt<0>();   // the zero-fill pieces past the last stage: the ring must be quiet before the next item's first DMA
    } else {
      if (s0 < s1) stage(0, s0);
      for (int st = s0; st < s1; ++st) {
        wait_vmcnt<0>();
        __syncthreads();
        if (st + 1 < s1) stage((st + 1 - s0) & 1, st + 1);
        const char *as = smem + ((st - s0) & 1) * 2 * OPB, *bs = as + OPB;
        kblock(as, bs);
        kblock(as + 32 * KROWB, bs + 32 * KROWB);
      }
    }
    // the next ticket rides on the barrier that ends the K loop; two slots in turn, so the slot written now was last read
    // two barriers ago (no barrier of its own: one behind the epilogue would have every wave wait for the slowest one's
    // store issue before the next item's first DMA)
    if (a.ctr && tid == 0) s_next[par] = nxt;
    __syncthreads();   // the ring is free again before the next item's first DMA
    int item_next;
    if (a.ctr) item_next = __builtin_amdgcn_readfirstlane(s_next[par]);   // wave-uniform: keeps the item's tile arithmetic scalar
    else if (item + a.cpg < n_full) item_next = item + a.cpg;
    else item_next = (item < n_full && j < n_items - n_full) ? n_full + j : n_items;

    // D rows = n (fg*4 + r within subtile jn), D cols = m (frow within subtile im)
    float *out;
    int64_t ldo;
    bool bounded;
    if (partial) { out = a.ws + (int64_t)slot * TN3_SLOT; ldo = 256; bounded = false; }
    else { out = dW + (int64_t)e * M * N + (int64_t)m0 * N + n0; ldo = N; bounded = true; }
#pragma unroll
    for (int jn = 0; jn < 4; ++jn)
#pragma unroll
      for (int im = 0; im < 8; ++im) {
        const int m = wm * 128 + im * 16 + frow, n = wn * 64 + jn * 16 + fg * 4;
        if (!bounded || (m0 + m < M && n0 + n < N)) *reinterpret_cast<f32x4 *>(out + (int64_t)m * ldo + n) = acc[jn][im];
      }
    if (want_bias && fg == 0) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int m = wm * 128 + (2 * wn + ii) * 16 + frow;
        if (partial) a.ws[(int64_t)slot * TN3_SLOT + 256 * 256 + m] = accb[ii][0];
        else if (m0 + m < M) dbias[(int64_t)e * M + m0 + m] = accb[ii][0];
      }
    }
    item = item_next;
  }
}

// sums the row-slices of the split tiles in slice order.  grid = (groups * max(1, cpg/2), 16)
__global__ void __launch_bounds__(256) tn3_fold_k(Tn3Args a) {
  const int half = max(1, a.cpg / 2);
  const int g = blockIdx.x / half, ri = blockIdx.x - g * half;
  const bool second = g >= a.E;
  const int e = second ? g - a.E : g;
  const Tn3Problem &pp = second ? a.p1 : a.p0;
  const Tn3Sched sc = tn3_sched(pp.m_tiles, pp.n_tiles, a.cpg);
  if (ri >= sc.rem || sc.s <= 1) return;
  const int tile = sc.full * a.cpg + ri;
  int mt, nt;
  tn3_tile_coord(tile, pp.m_tiles, pp.n_tiles, mt, nt);
  const int m0 = mt * 256, n0 = nt * 256;
  const float *src = a.ws + (int64_t)(g * a.cpg + ri * sc.s) * TN3_SLOT;
  float *out = pp.dW + (int64_t)e * pp.M * pp.N;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = (blockIdx.y * 1024 + q * 256 + threadIdx.x) * 4;
    const int m = idx >> 8, n = idx & 255;
    if (m0 + m >= pp.M || n0 + n >= pp.N) continue;
    float4 sum = *reinterpret_cast<const float4 *>(src + idx);
    for (int sl = 1; sl < sc.s; ++sl) {
      const float4 v = *reinterpret_cast<const float4 *>(src + (int64_t)sl * TN3_SLOT + idx);
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    *reinterpret_cast<float4 *>(out + (int64_t)(m0 + m) * pp.N + n0 + n) = sum;
  }
  if (pp.dbias && nt == 0 && blockIdx.y == 0 && m0 + (int)threadIdx.x < pp.M) {
    float sum = src[256 * 256 + threadIdx.x];
    for (int sl = 1; sl < sc.s; ++sl) sum += src[(int64_t)sl * TN3_SLOT + 256 * 256 + threadIdx.x];
    pp.dbias[(int64_t)e * pp.M + m0 + threadIdx.x] = sum;
  }
}

// ------------------------------------------------------------------------------------------
// TN v5 (bf16): 256 x 352 / 352 x 256 output tiles for weight gradients with a 704-wide side (the H = 704 family:
// dW1 [2816, 704], dW2 [704, 2816]).  On 256 x 256 tiles 704 is 256 + 256 + 192: 33 tiles per (problem, expert), a quarter
// of every third tile's MFMAs multiply zeros (8.3 % of the work) and every 32-deep interval carries the same barrier /
// fragment-read / DMA-issue overhead under 32 MFMAs per wave.  Here an interval has 44 MFMAs per wave under the same
// overhead, a group has 22 tiles, and a tile moves 17 % fewer operand bytes per flop.  Scheme of v3 (CUs dealt to the
// (problem, expert) groups, lock-step rounds, row-split remainder tiles folded in slice order, optional item queue,
// bias gradient from the matrix pipe); what differs:
//  - the 352-wide operand's k-rows (704 B) sit on a 768-byte LDS pitch, so the 32-byte window swizzle f(k) of v3 stays
//    inside the row (windows 16..21 map into 16..23) and the bank pattern of the transposed reads is v3's;
//  - its DMA pieces (1 KiB of LDS = 1 1/3 rows) take a per-lane source offset from a table of three (the pattern
//    repeats every three pieces = four rows); pad positions repeat the row's last chunk;
//  - 8 waves as 2 (M) x 4 (N) of 176 x 64 (WIDE_M) or 4 (M) x 2 (N) of 64 x 176: 176 accumulator registers, so the
//    eleven fragments of the wide side are read two ahead of the MFMAs that consume them (as grouped_gemm_nt352p_k
//    does; v3's half-interval stagger would have to hold all fifteen fragments across the barrier);
//  - three slots of 32 k-rows (40 KiB each), two stages in flight, the five DMA pieces of a wave per interval issued
//    one by one between the MFMA groups (see v3).
// ------------------------------------------------------------------------------------------
constexpr int TN5_PW = 768;                         // LDS pitch of a 352-wide k-row
constexpr int TN5_STAGE = 32 * (TN5_PW + 512);      // A image then B image
constexpr int TN5_SLOT = 256 * 352 + 352;           // floats per partial tile (+ its partial bias sums)
struct Tn5Args {
  Tn3Problem p0, p1;   // (m_tiles / n_tiles in units of this kernel's tiles)
  int wide_m0, wide_m1;
  int slice_major;
  int nprob, E, cpg;
  float *ws;
  int *ctr;
};

template <bool WIDE_M>
__device__ __forceinline__ void tn5_body(const Tn5Args &a, const int32_t *__restrict__ offsets, char *smem, int g, int j,
                                         bool second) {
  typedef bf16_t T;
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  constexpr int IM = WIDE_M ? 11 : 4, JN = WIDE_M ? 4 : 11;       // 16-wide subtiles of a wave along m / n
  constexpr int TMx = WIDE_M ? 352 : 256, TNx = WIDE_M ? 256 : 352;
  constexpr int PA = WIDE_M ? TN5_PW : 512, PB = WIDE_M ? 512 : TN5_PW;
  constexpr int OPA = 32 * PA;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = WIDE_M ? wave >> 2 : wave >> 1, wn = WIDE_M ? wave & 3 : wave & 1;
  const int e = second ? g - a.E : g;
  const T *__restrict__ A = second ? a.p1.A : a.p0.A, *__restrict__ Bm = second ? a.p1.Bm : a.p0.Bm;
  float *__restrict__ dW = second ? a.p1.dW : a.p0.dW, *__restrict__ dbias = second ? a.p1.dbias : a.p0.dbias;
  const int M = second ? a.p1.M : a.p0.M, N = second ? a.p1.N : a.p0.N;
  const int m_tiles = second ? a.p1.m_tiles : a.p0.m_tiles, n_tiles = second ? a.p1.n_tiles : a.p0.n_tiles;
  const Tn3Sched sc = tn3_sched(m_tiles, n_tiles, a.cpg);
  const int r_begin = offsets[e], rows = offsets[e + 1] - r_begin;
  const int nsteps = (rows + 63) / 64;
  const int ldA = M * (int)sizeof(T), ldB = N * (int)sizeof(T);

  // per-lane source offsets of the DMA pieces.  512-byte rows: piece p = two k-rows (v3's scheme).  768-pitch rows: lane i
  // of piece p fills LDS byte 1024 p + 16 i = row r, position pos; it fetches chunk pos ^ (f(r) << 1) of that row.
  const int hi = lane >> 5;
  const int nchunk = (lane & 31) ^ (hi << 1) ^ ((wave & 1) << 2) ^ (((wave >> 2) & 1) << 3);
  const int ld_narrow = WIDE_M ? ldB : ldA, ld_wide = WIDE_M ? ldA : ldB;
  const uint32_t vn0 = (uint32_t)(hi * ld_narrow + nchunk * 16);
  uint32_t vw[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int o = 1024 * (wave + 8 * q) + 16 * lane;
    const int r = o / TN5_PW, pos = (o - r * TN5_PW) >> 4;
    const int fk = (r & 3) | (((r >> 3) & 1) << 2);
    const int c = min(pos ^ (fk << 1), 43);
    vw[q] = (uint32_t)r * (uint32_t)ld_wide + (uint32_t)(c * 16);
  }
  const int frow = lane & 15, fg = lane >> 4;
  const int trow = frow >> 2, tcol4 = (frow & 3) * 4;
  const int lane_sw = trow | ((fg & 1) << 2);
  const int rd_a = (8 * fg + trow) * PA + tcol4 * 2, rd_b = (8 * fg + trow) * PB + tcol4 * 2;
  const int wn_u = wn;

  int *s_next = reinterpret_cast<int *>(smem + 3 * TN5_STAGE);
  const int n_full = sc.full * a.cpg, n_items = n_full + (sc.rem ? sc.rem * sc.s : 0);
  int item = j;
  if (a.ctr) {   // (queue mode: the first item from the group's counter too - see grouped_gemm_tn3_k)
    if (tid == 0) s_next[0] = atomicAdd(a.ctr + g * TN3_CTR_STRIDE, 1);
    __syncthreads();
    item = __builtin_amdgcn_readfirstlane(s_next[0]);
  }
  for (int par = 1; item < n_items; par ^= 1) {
    int nxt = 0;
    if (a.ctr && tid == 0) nxt = atomicAdd(a.ctr + g * TN3_CTR_STRIDE, 1);
    int tile, s0 = 0, s1 = nsteps, slot = 0;
    bool partial = false;
    if (item < n_full) {
      tile = item;
    } else {
      const int r = item - n_full;
      // which (tile, row slice) item r is.  Several groups: tile-major (a tile's slices on neighbouring work-groups).  ONE group
      // holding all the CUs (a dense layer: slice_major): slice-major - the work-groups that share a row slice, and with it
      // one operand's rows, are neighbours, i.e. on one XCD and its L2 (tile-major they sat cpg / rem apart: every tile's
      // column of slices fetched that operand from HBM again)
      int ri, sl;
      if (a.slice_major) { sl = r / sc.rem; ri = r - sl * sc.rem; } else { ri = r / sc.s; sl = r - ri * sc.s; }
      const int per = (nsteps + sc.s - 1) / sc.s;
      tile = n_full + ri;
      s0 = min(sl * per, nsteps);
      s1 = min(s0 + per, nsteps);
      partial = sc.s > 1;
      slot = g * a.cpg + ri * sc.s + sl;
    }
    int mt, nt;
    tn3_tile_coord(tile, m_tiles, n_tiles, mt, nt);
    const int m0 = mt * TMx, n0 = nt * TNx;
    const bool want_bias = dbias != nullptr && nt == 0;
    const v4i ars = raw_buffer_rsrc(A + (int64_t)r_begin * M + m0, rows > 0 ? (uint32_t)(rows * ldA - m0 * (int)sizeof(T)) : 0u);
    const v4i brs = raw_buffer_rsrc(Bm + (int64_t)r_begin * N + n0, rows > 0 ? (uint32_t)(rows * ldB - n0 * (int)sizeof(T)) : 0u);
    const v4i &wrs = WIDE_M ? ars : brs, &nrs = WIDE_M ? brs : ars;
    const uint32_t lds0 = lds_addr_of(smem);
    const int nsub = 2 * (s1 - s0);

    f32x4 acc[JN][IM];
#pragma unroll
    for (int jn = 0; jn < JN; ++jn)
#pragma unroll
      for (int im = 0; im < IM; ++im) acc[jn][im] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int NB = WIDE_M ? 3 : 2;   // m-subtiles whose bias sums this wave forms
    f32x4 accb[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) accb[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto mma_ones = [&](f32x4 &d, const bf16x8 &b) {   // (the all-ones fragment is rebuilt where it is used: four registers less)
      bf16x8 ones;
#pragma unroll
      for (int q = 0; q < 8; ++q) ones[q] = (bf16_t)1.0f;
      mma(d, ones, b);
    };

    // piece q (0..4) of this wave's share of stage u: 0..2 the 768-pitch image, 3..4 the 512-pitch image.  Past the item's
    // last stage the piece is still issued, from beyond the descriptor's range: zeros into a slot nobody reads again
    auto piece = [&](int u, uint32_t slot_off, int q) {   // slot_off = (u % 3) * TN5_STAGE, kept as a running value by the caller
      const uint32_t wimg = lds0 + slot_off + (WIDE_M ? 0 : OPA), nimg = lds0 + slot_off + (WIDE_M ? OPA : 0);
      const uint32_t row0 = (uint32_t)(s0 * 64 + u * 32);
      const bool live = u < nsub;
      if (q < 3) lds_dma16(wrs, wimg + (wave + 8 * q) * 1024, live ? vw[q] + row0 * (uint32_t)ld_wide : 0xfffffff0u);
      else {
        const int p = (q - 3) * 8 + wave;
        lds_dma16(nrs, nimg + p * 1024, live ? vn0 + (row0 + 2 * p) * (uint32_t)ld_narrow : 0xfffffff0u);
      }
    };
    // (the swizzle key is laundered once per interval: the fifteen window offsets are loop invariants that hipcc would
    // otherwise keep in registers across the K loop - and spill: 200-270 B of scratch, reloaded every interval)
    int lsw = lane_sw;
    auto read_a = [&](const char *as, int im) {
      union { bf16x8 v; s16x4 h[2]; } u;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
        u.h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(
            as + rd_a + (((wm * IM + im) ^ lsw) << 5) + (4 * hh) * PA));
      return u.v;
    };
    auto read_b = [&](const char *bs, int jn) {
      union { bf16x8 v; s16x4 h[2]; } u;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh)
        u.h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(
            bs + rd_b + (((wn * JN + jn) ^ lsw) << 5) + (4 * hh) * PB));
      return u.v;
    };
#pragma unroll
    for (int q = 0; q < 5; ++q) piece(0, 0u, q);
#pragma unroll
    for (int q = 0; q < 5; ++q) piece(1, (uint32_t)TN5_STAGE, q);
    uint32_t off_cur = 0u, off_nxt = TN5_STAGE, off_fill = 2 * TN5_STAGE;   // slots of stages u, u + 1, u + 2
    for (int u = 0; u < nsub; ++u) {
      wait_vmcnt<5>();   // stage u has landed once only stage u + 1's five pieces are outstanding (vmcnt retires in order)
      lds_barrier();     // ... for every wave, and nobody still reads slot (u - 1) % 3, which stage u + 2 overwrites
      const char *as = smem + off_cur, *bs = as + OPA;
      asm volatile("" : "+v"(lsw));
      bf16x8 af[IM], bf[JN];
      __builtin_amdgcn_s_setprio(1);
      if constexpr (WIDE_M) {
#pragma unroll
        for (int jn = 0; jn < JN; ++jn) bf[jn] = read_b(bs, jn);
        af[0] = read_a(as, 0); af[1] = read_a(as, 1);
#pragma unroll
        for (int im = 0; im < IM; ++im) {
          if (im + 2 < IM) af[im + 2] = read_a(as, im + 2);
          __builtin_amdgcn_sched_barrier(0);   // keeps the reads from being hoisted into one block of 44 live registers
#pragma unroll
          for (int jn = 0; jn < JN; ++jn) mma(acc[jn][im], bf[jn], af[im]);
          if (want_bias && (im & 3) == wn_u) mma_ones(accb[im >> 2], af[im]);   // (wave-uniform: MFMA ignores EXEC)
          if (!(im & 1) && im < 10) {
            __builtin_amdgcn_sched_barrier(0);
            piece(u + 2, off_fill, im >> 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else {
#pragma unroll
        for (int im = 0; im < IM; ++im) af[im] = read_a(as, im);
        bf[0] = read_b(bs, 0); bf[1] = read_b(bs, 1);
#pragma unroll
        for (int jn = 0; jn < JN; ++jn) {
          if (jn + 2 < JN) bf[jn + 2] = read_b(bs, jn + 2);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int im = 0; im < IM; ++im) mma(acc[jn][im], bf[jn], af[im]);
          if (jn == 0 && want_bias) {
#pragma unroll
            for (int im = 0; im < IM; ++im)
              if ((im >> 1) == wn_u) mma_ones(accb[im & 1], af[im]);
          }
          if (!(jn & 1) && jn < 10) {
            __builtin_amdgcn_sched_barrier(0);
            piece(u + 2, off_fill, jn >> 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      __builtin_amdgcn_s_setprio(0);
      { const uint32_t t = off_cur; off_cur = off_nxt; off_nxt = off_fill; off_fill = t; }
    }
    wait_vmcnt<0>();   // the zero-fill pieces past the last stage: the ring must be quiet before the next item's first DMA
    if (a.ctr && tid == 0) s_next[par] = nxt;
    __syncthreads();
    int item_next;
    if (a.ctr) item_next = __builtin_amdgcn_readfirstlane(s_next[par]);
    else if (item + a.cpg < n_full) item_next = item + a.cpg;
    else item_next = (item < n_full && j < n_items - n_full) ? n_full + j : n_items;

    // D rows = n (fg*4 + r within subtile jn), D cols = m (frow within subtile im)
    float *out;
    int64_t ldo;
    bool bounded;
    if (partial) { out = a.ws + (int64_t)slot * TN5_SLOT; ldo = TNx; bounded = false; }
    else { out = dW + (int64_t)e * M * N + (int64_t)m0 * N + n0; ldo = N; bounded = true; }
#pragma unroll
    for (int jn = 0; jn < JN; ++jn)
#pragma unroll
      for (int im = 0; im < IM; ++im) {
        const int m = (wm * IM + im) * 16 + frow, n = (wn * JN + jn) * 16 + fg * 4;
        if (!bounded || (m0 + m < M && n0 + n < N)) *reinterpret_cast<f32x4 *>(out + (int64_t)m * ldo + n) = acc[jn][im];
      }
    if (want_bias && fg == 0) {
#pragma unroll
      for (int im = 0; im < IM; ++im) {
        const bool mine = WIDE_M ? (im & 3) == wn_u : (im >> 1) == wn_u;
        if (mine) {
          const int m = (wm * IM + im) * 16 + frow;
          const float v = accb[WIDE_M ? im >> 2 : im & 1][0];
          if (partial) a.ws[(int64_t)slot * TN5_SLOT + TMx * TNx + m] = v;
          else if (m0 + m < M) dbias[(int64_t)e * M + m0 + m] = v;
        }
      }
    }
    item = item_next;
  }
}

__global__ void __launch_bounds__(NT2)
grouped_gemm_tn5_k(Tn5Args a, const int32_t *__restrict__ offsets) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int g = id / a.cpg, j = id - g * a.cpg;
  const bool second = g >= a.E;
  if (second ? a.wide_m1 : a.wide_m0) tn5_body<true>(a, offsets, smem, g, j, second);
  else tn5_body<false>(a, offsets, smem, g, j, second);
}

// sums the row-slices of the split tiles in slice order.  grid = (groups * max(1, cpg/2), 88 / cpb): cpb chunks of 1024 floats
// per work-group - four where a tile has a few slices (the expert groups: two), ONE where it has many (a single group holding
// all the CUs: 85 slices per tile and only `rem` tiles - with four chunks each that fold ran on 66 work-groups)
__global__ void __launch_bounds__(256) tn5_fold_k(Tn5Args a, int cpb) {
  const int half = max(1, a.cpg / 2);
  const int g = blockIdx.x / half, ri = blockIdx.x - g * half;
  const bool second = g >= a.E;
  const int e = second ? g - a.E : g;
  const Tn3Problem &pp = second ? a.p1 : a.p0;
  const bool wide_m = second ? a.wide_m1 : a.wide_m0;
  const int TMx = wide_m ? 352 : 256, TNx = wide_m ? 256 : 352;
  const Tn3Sched sc = tn3_sched(pp.m_tiles, pp.n_tiles, a.cpg);
  if (ri >= sc.rem || sc.s <= 1) return;
  const int tile = sc.full * a.cpg + ri;
  int mt, nt;
  tn3_tile_coord(tile, pp.m_tiles, pp.n_tiles, mt, nt);
  const int m0 = mt * TMx, n0 = nt * TNx;
  const float *src = a.ws + (int64_t)(g * a.cpg + ri * sc.s) * TN5_SLOT;
  float *out = pp.dW + (int64_t)e * pp.M * pp.N;
  for (int q = 0; q < cpb; ++q) {
    const int idx = ((blockIdx.y * cpb + q) * 256 + threadIdx.x) * 4;   // < 256 * 352 = 88 * 1024
    const int m = idx / TNx, n = idx - m * TNx;
    if (m0 + m >= pp.M || n0 + n >= pp.N) continue;
    float4 sum = *reinterpret_cast<const float4 *>(src + idx);
#pragma unroll 4
    for (int sl = 1; sl < sc.s; ++sl) {
      const float4 v = *reinterpret_cast<const float4 *>(src + (int64_t)sl * TN5_SLOT + idx);
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    *reinterpret_cast<float4 *>(out + (int64_t)(m0 + m) * pp.N + n0 + n) = sum;
  }
  if (pp.dbias && nt == 0 && blockIdx.y == 0) {
    for (int m = threadIdx.x; m < TMx; m += 256) {
      if (m0 + m >= pp.M) continue;
      float sum = src[TMx * TNx + m];
      for (int sl = 1; sl < sc.s; ++sl) sum += src[(int64_t)sl * TN5_SLOT + TMx * TNx + m];
      pp.dbias[(int64_t)e * pp.M + m0 + m] = sum;
    }
  }
}

int device_cu_count() {
  static const int ncu = [] {   // queried once: hipGetDeviceProperties costs ~1 ms of host time per call
    int n = 256, dev_id = 0;
    if (hipGetDevice(&dev_id) == hipSuccess) hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev_id);
    return n > 0 ? n : 256;
  }();
  return ncu;
}

// returns APERTIS_ERR_UNSUPPORTED when the shape does not fit v3's assumptions (the caller falls back to v2)
int launch_tn3(const Tn3Problem &q0, const Tn3Problem *q1, int64_t E, int64_t max_rows, const int32_t *offsets, float *ws,
               int64_t ws_bytes, bool item_queue, hipStream_t st) {
  const int nprob = q1 ? 2 : 1;
  const int64_t groups = nprob * E;
  const int ncu = device_cu_count();
  if (!ws || groups > ncu) return APERTIS_ERR_UNSUPPORTED;
  // (short groups - under ~2048 rows each - do not amortise a 256-row-deep slice per CU plus the fold: callers leave
  // `ws` NULL for those and get the 128 x 128 kernel; the choice is the caller's, the library has no hidden switch)
  const int cpg = (int)(ncu / groups);
  const int grid = (int)(groups * cpg);
  const int64_t slots_bytes = (int64_t)grid * TN3_SLOT * (int64_t)sizeof(float);
  if (ws_bytes < slots_bytes + TN3_CTR_BYTES || (((uintptr_t)ws) & 15)) return APERTIS_ERR_UNSUPPORTED;
  const int64_t ldmax = std::max<int64_t>(std::max(q0.M, q0.N), q1 ? std::max(q1->M, q1->N) : 0) * 2;
  if ((max_rows + 256) * ldmax >= 0xffffffffLL) return APERTIS_ERR_UNSUPPORTED;   // 32-bit buffer offsets
  Tn3Args a;
  a.p0 = q0;
  a.p1 = q1 ? *q1 : Tn3Problem{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
  a.nprob = nprob; a.E = (int)E; a.cpg = cpg; a.ws = ws;
  a.ctr = item_queue ? reinterpret_cast<int *>(reinterpret_cast<char *>(ws) + slots_bytes) : nullptr;
  if (a.ctr && hipMemsetAsync(a.ctr, 0, (size_t)groups * TN3_CTR_STRIDE * sizeof(int), st) != hipSuccess) return APERTIS_ERR_LAUNCH;
  const size_t lds = 4 * 64 * 512 + 16;
  int ring = 2;
  auto k3 = ring == 2 ? grouped_gemm_tn3_k<true, true> : ring == 1 ? grouped_gemm_tn3_k<true, false> : grouped_gemm_tn3_k<false, false>;
  hipFuncSetAttribute((const void *)k3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k3, dim3((unsigned)grid), dim3(NT2), lds, st, a, offsets);
  bool split = false;
  for (int q = 0; q < nprob; ++q) {
    const Tn3Problem &pp = q ? a.p1 : a.p0;
    const Tn3Sched sc = tn3_sched(pp.m_tiles, pp.n_tiles, cpg);
    split |= sc.rem && sc.s > 1;
  }
  if (split) hipLaunchKernelGGL(tn3_fold_k, dim3((unsigned)(groups * std::max(1, cpg / 2)), 16), dim3(256), 0, st, a);
  return apertis_check_launch();
}

// which tile shape of v5 suits an [M, N] weight gradient: 1 = 352 x 256, 0 = 256 x 352, -1 = neither (v3's 256 x 256
// tiles waste no more than 5 % less)
int tn5_variant(int64_t M, int64_t N) {
  auto area = [&](int64_t tm, int64_t tn) { return ceil_div64(M, tm) * ceil_div64(N, tn) * tm * tn; };
  const int64_t a3 = area(256, 256), aw = area(352, 256), an = area(256, 352);
  if (M < 256 || N < 256 || std::min(aw, an) * 100 > a3 * 95) return -1;
  return aw < an ? 1 : 0;
}

// returns APERTIS_ERR_UNSUPPORTED when the shapes do not suit v5 (the caller falls back to v3)
// (q1p == nullptr: one problem - its groups get all the CUs; with E = 1 that is a dense weight gradient as a row-split GEMM)
int launch_tn5(Tn3Problem q0, const Tn3Problem *q1p, int64_t E, int64_t max_rows, const int32_t *offsets, float *ws, int64_t ws_bytes,
               bool item_queue, hipStream_t st, int force_v0 = -1) {
  Tn3Problem q1 = q1p ? *q1p : q0;
  const int nprob = q1p ? 2 : 1;
  const int v0 = force_v0 >= 0 ? force_v0 : tn5_variant(q0.M, q0.N), v1 = q1p ? tn5_variant(q1.M, q1.N) : v0;
  const int64_t groups = nprob * E;
  const int ncu = device_cu_count();
  if (!ws || v0 < 0 || v1 < 0 || groups > ncu) return APERTIS_ERR_UNSUPPORTED;
  const int cpg = (int)(ncu / groups);
  const int grid = (int)(groups * cpg);
  const int64_t slots_bytes = (int64_t)grid * TN5_SLOT * (int64_t)sizeof(float);
  if (ws_bytes < slots_bytes + TN3_CTR_BYTES || (((uintptr_t)ws) & 15)) return APERTIS_ERR_UNSUPPORTED;
  const int64_t ldmax = std::max<int64_t>(std::max(q0.M, q0.N), std::max(q1.M, q1.N)) * 2;
  if ((max_rows + 256) * ldmax >= 0xffffffffLL) return APERTIS_ERR_UNSUPPORTED;   // 32-bit buffer offsets
  q0.m_tiles = (int)ceil_div64(q0.M, v0 ? 352 : 256); q0.n_tiles = (int)ceil_div64(q0.N, v0 ? 256 : 352);
  q1.m_tiles = (int)ceil_div64(q1.M, v1 ? 352 : 256); q1.n_tiles = (int)ceil_div64(q1.N, v1 ? 256 : 352);
  Tn5Args a;
  a.p0 = q0; a.p1 = q1; a.wide_m0 = v0; a.wide_m1 = v1;
  a.nprob = nprob; a.E = (int)E; a.cpg = cpg; a.ws = ws;
  a.slice_major = groups == 1;
  a.ctr = item_queue ? reinterpret_cast<int *>(reinterpret_cast<char *>(ws) + slots_bytes) : nullptr;
  if (a.ctr && hipMemsetAsync(a.ctr, 0, (size_t)groups * TN3_CTR_STRIDE * sizeof(int), st) != hipSuccess) return APERTIS_ERR_LAUNCH;
  const size_t lds = 3 * TN5_STAGE + 16;
  hipFuncSetAttribute((const void *)grouped_gemm_tn5_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(grouped_gemm_tn5_k, dim3((unsigned)grid), dim3(NT2), lds, st, a, offsets);
  bool split = false;
  for (int q = 0; q < nprob; ++q) {
    const Tn3Problem &pp = q ? a.p1 : a.p0;
    const Tn3Sched sc = tn3_sched(pp.m_tiles, pp.n_tiles, cpg);
    split |= sc.rem && sc.s > 1;
  }
  if (split) {
    const int cpb = groups == 1 ? 1 : 4;
    hipLaunchKernelGGL(tn5_fold_k, dim3((unsigned)(groups * std::max(1, cpg / 2)), 88 / cpb), dim3(256), 0, st, a, cpb);
  }
  return apertis_check_launch();
}

// (behind every other kernel of this file: the kernels in front keep their places in the code object)
#include "gemm_nt2i.h"   // grouped_gemm_nt2i_k: the saved-gradient forward with its epilogue inside the next tile's K loop (round 6)

template <typename T> bool aligned16(const void *p, int64_t ld) {
  return (((uintptr_t)p) & 15) == 0 && ((ld * sizeof(T)) & 15) == 0;
}

template <typename T, typename TO>
int launch_nt(const void *A, const void *W, const float *bias, const int32_t *offsets, void *C, void *pre_act,
              const void *mul_pre, int64_t max_rows, int64_t N, int64_t K, int64_t ldw, int64_t E, int act, float drop_p,
              uint64_t seed, int32_t *tile_queue, hipStream_t st) {
  if (!aligned16<T>(A, K) || !aligned16<T>(W, ldw) || !aligned16<TO>(C, N) || (pre_act && !aligned16<TO>(pre_act, N)) ||
      (mul_pre && !aligned16<TO>(mul_pre, N)))
    return APERTIS_ERR_UNSUPPORTED;
  // the saved-gradient forms of the epilogue exist in the two-per-CU kernel only
  const int act_flags = act;
  const bool flagged = (act & (APERTIS_ACT_SAVE_GRAD | APERTIS_ACT_MUL_SAVED)) != 0;
  act &= 0xff;
  if (flagged && ((act_flags & APERTIS_ACT_SAVE_GRAD) ? !pre_act : !mul_pre)) return APERTIS_ERR_ARG;
  if ((act_flags & APERTIS_ACT_SAVE_GRAD) && (act != APERTIS_ACT_GELU || (max_rows + 256) * N >= 0x100000000LL)) return APERTIS_ERR_UNSUPPORTED;
  const int n_tiles = (int)ceil_div64(N, BN);
  const int64_t m_tiles = ceil_div64(max_rows, BM) + E;  // each group adds at most one partial tile
  const int64_t grid = m_tiles * n_tiles;
  if (grid > 0x7fffffffLL) return APERTIS_ERR_UNSUPPORTED;
  if constexpr (sizeof(T) == 2 && sizeof(TO) == 2) {
    // a handful of rows (the decode step): a wave per 16 output columns, operands straight from global memory
    if (max_rows <= 64 && !flagged && !pre_act && !mul_pre && drop_p <= 0.f && K % 8 == 0 && N % 4 == 0 && ldw % 8 == 0 && E <= 65535 &&
        ceil_div64(N, 16) <= 0x7fffffff) {
      if (K >= 2048 && SKINNY_LONG_K_WAVES == 16)  // (sixteen waves: a wave's share of K = 2816 is six 32-deep steps = ONE batch in flight instead of three dependent ones)
        hipLaunchKernelGGL((grouped_gemm_nt_skinny_k<TO, 16>), dim3((unsigned)ceil_div64(N, 16), (unsigned)E), dim3(1024), 0, st,
                           (const bf16_t *)A, (const bf16_t *)W, bias, offsets, (TO *)C, (int)N, (int)K, (int)ldw, act, (int)max_rows);
      else if (K >= 512)   // (a wave's share is then one batch of eight 32-deep steps or a few: the K walk is a chain of round trips)
        hipLaunchKernelGGL((grouped_gemm_nt_skinny_k<TO, 4>), dim3((unsigned)ceil_div64(N, 16), (unsigned)E), dim3(256), 0, st,
                           (const bf16_t *)A, (const bf16_t *)W, bias, offsets, (TO *)C, (int)N, (int)K, (int)ldw, act, (int)max_rows);
      else
        hipLaunchKernelGGL((grouped_gemm_nt_skinny_k<TO, 1>), dim3((unsigned)ceil_div64(N, 16), (unsigned)E), dim3(64), 0, st,
                           (const bf16_t *)A, (const bf16_t *)W, bias, offsets, (TO *)C, (int)N, (int)K, (int)ldw, act, (int)max_rows);
      return apertis_check_launch();
    }
    // the 256x256 kernel steps K in 64s: K itself may be ragged when W's rows are zero-padded to the step.
    // One group (dense projection) takes it at any width: narrow outputs are HBM-bound and the
    // persistent kernel's cross-tile prefetch matters more than the MFMA work a partial n-tile wastes
    const bool kpad_ok = K % 64 == 0 || ldw >= ceil_div64(K, 64) * 64;
    // two work-groups per CU pay off when the epilogue is heavy next to the K loop (activation / dropout /
    // second output on a short K); long K loops run faster on the 256 x 256 tile (fewer operand bytes per flop).
    // ... and for the SSM block's dense projections (one group, short K, HBM-bound): 92 vs 114 us at N=352/K=704, 72 vs 89
    // at N=704/K=176, 56 vs 64 at N=400/K=176; the N=176 data gradients stay on the 256-wide tile (63 vs 60 us)
    const bool use2x = ((act != APERTIS_ACT_NONE || drop_p > 0.f || pre_act || mul_pre) && K <= 1024 && N >= 512) ||
                       // (one 352-wide n-tile reads X once: 114 vs 121 us on the SSM input projection, N = 352, K = 704)
                       (E == 1 && K <= 1024 && N >= 256 && !(N == BN5 && K % 64 == 0 && ldw == K)) ||
                       // narrow expert outputs (the H = 256 family's fc2 forward / fc1 data gradient, N = 256, K = 1024):
                       // 67 us here, 77 on the 256 x 256 tile, 90 on the 128 x 128 kernel they used to fall to
                       (E > 1 && K <= 1024 && N >= 256 && N < 512) ||
                       // narrow dense outputs (the H = 256 family's x_param / out_proj data gradients, N = 64): one half-empty
                       // 128-wide n-tile of this kernel instead of the 128 x 128 register-staged kernel (29 us for 42 MB there)
                       (E == 1 && K <= 1024 && N >= 64 && N < 128);
    // outputs a multiple of 352 wide with a plain epilogue: the 256 x 352 tile (two passes over X for N = 704 instead of three)
    const bool use352 = !use2x && act == APERTIS_ACT_NONE && drop_p <= 0.f && !pre_act && !mul_pre && N % BN5 == 0 && K % 64 == 0 &&
                        ldw == K && K >= 128 && max_rows >= 4096 && E <= 1024;
    if (use352) {
      const int nt5 = (int)(N / BN5);
      const int64_t grid5 = (ceil_div64(max_rows, BM2) + E) * nt5;
      if (grid5 < 0x7fffffffLL) {
        const int gp = (int)std::min<int64_t>(grid5, device_cu_count());   // one persistent work-group per CU
        const size_t lds5 = 2 * BUF5 + 4096 + 16;                           // ring + group offsets
        constexpr int solo = 4;   // (re-measured with the spread fills, round 3: 0..6 all within noise)
        if (tile_queue && hipMemsetAsync(tile_queue, 0, NTQ_INTS * sizeof(int32_t), st) != hipSuccess) return APERTIS_ERR_LAUNCH;
        auto k5 = grouped_gemm_nt352p_k<TO>;
        hipFuncSetAttribute((const void *)k5, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5);
        hipLaunchKernelGGL(k5, dim3((unsigned)gp), dim3(NT2), lds5, st, (const bf16_t *)A, (const bf16_t *)W, bias, offsets, (TO *)C,
                           (int)N, (int)K, (int)E, nt5, (int)grid5, solo, (int *)tile_queue);
        return apertis_check_launch();
      }
    }
    // the persistent 256 x 256 tile on a four-slot ring (grouped_gemm_nt4r_k): every epilogue form of the two-per-CU kernel
    const bool ragged2x = K % 32 != 0;
    // Taken for the calls with a heavy epilogue on a short K (the expert fc1 forward and the fused fc2 data gradient; the dense
    // FFN's as one group).  Same box, best of 5 cold runs, two-per-CU kernel -> this one (tools/probes/gemm_probe.hip, PROBE_R4):
    //   H=704 I=2816 B=32: saved-gradient forward 1195-1200 -> 1096-1105 us, fused data gradient 949-951 -> 792-797
    //   H=704 I=2816 B=44: 1365 -> 1324-1349, 1116 -> 997-1006 (another box)
    //   H=256 I=1024 B=16: 193 -> 154-166, 181 -> 108            H=896 I=3584: 820-882 -> 749-757, 715-725 -> 612-628
    // Plain epilogues stay where they were (N=1024, K=256: 78 us there, 88 here; N=3584, K=896: 565 there, 508 here - not routed).
    // A caller that passes a tile queue (data-parallel runs: a collective may hold CUs) keeps the non-persistent kernel.
    const bool use4r = ((act != APERTIS_ACT_NONE || drop_p > 0.f || pre_act || mul_pre) && K <= 1024 && N >= 512) ||
                       // ... and for the SSM block's dense projections (one group, short K: byte- and latency-bound - the stage
                       // stream through tile boundaries hides a tile's prologue and epilogue, and 256-wide n-tiles read X half as
                       // often).  Same box, B = 44 (`gemm_probe 44 704 2816 dense`), two-per-CU -> this kernel: N=704 K=176 121 ->
                       // 102 us, N=704 K=352 162 -> 141, N=448 K=176 83 -> 74, N=176 K=448 63-67 -> 60, N=176 K=704 85-92 -> 83;
                       // N=352 K=704 stays on the 352-wide tile (107-110 there, 111 here)
                       (E == 1 && K <= 1024 && N >= 128 && !(N == BN5 && K % 64 == 0 && ldw == K));
    // the saved-gradient forward (GELU; the expert fc1 forward) with K >= 512: one wave per SIMD, the epilogue of tile i inside
    // the K loop of tile i + 1 (grouped_gemm_nt2i_k, gemm_nt2i.h).  Not under a tile queue (data-parallel steps: the ring kernel).
    // Bit-identical to the ring kernel and, at the bench shape, SLOWER (1678 against 1452 us: with one wave per SIMD the six
    // LDS-DMA pieces of a sub-step stall the only wave that could feed the matrix pipe - profiles/r6_probe_nt2i_vs_nt4r.log): taken
    // only when the caller asks for it (APERTIS_ACT_INTERLEAVED).
    if ((act_flags & APERTIS_ACT_INTERLEAVED) && (act_flags & APERTIS_ACT_SAVE_GRAD) && pre_act && !mul_pre && act == APERTIS_ACT_GELU && !tile_queue && K % 32 == 0 &&
        K >= 512 && ldw == K && N % 8 == 0 && N >= 512 && max_rows >= 4096 && E <= 1024) {
      const int nt5 = (int)ceil_div64(N, BN3);
      const int64_t grid5 = (ceil_div64(max_rows, BM3) + E) * nt5;
      if (grid5 < 0x7fffffffLL) {
        const int gp = (int)std::min<int64_t>(grid5, device_cu_count());
        const int lds5 = RING5I + 4 * 1024;   // ring + a bias area per wave
        auto k5 = drop_p > 0.f ? grouped_gemm_nt2i_k<TO, true> : grouped_gemm_nt2i_k<TO, false>;
        hipFuncSetAttribute((const void *)k5, hipFuncAttributeMaxDynamicSharedMemorySize, lds5);
        hipLaunchKernelGGL(k5, dim3((unsigned)gp), dim3(NT3), lds5, st, (const bf16_t *)A, (const bf16_t *)W, bias, offsets, (TO *)C,
                           (TO *)pre_act, (int)N, (int)K, (int)ldw, (int)E, nt5, (int)grid5, drop_p, seed, 4, 8);
        return apertis_check_launch();
      }
    }
    // (a caller's tile queue - data-parallel steps - is taken when the kernel's ticket hand-over fits: K >= 352 and a grid that
    //  is a multiple of the 8 XCDs; otherwise such calls stay on the queue-driven two-per-CU / 256 x 256 kernels below.  Round 6:
    //  the N > 1 step used to lose this kernel altogether - 956 against 902 us per expert NT call at B = 44)
    const int64_t grid4q = (ceil_div64(max_rows, BM4) + E) * ceil_div64(N, BN4);
    const bool queue4_ok = tile_queue && ceil_div64(K, 32) >= 11 && grid4q >= device_cu_count() && device_cu_count() % 8 == 0;
    if (use4r && !(mul_pre && bias) && (!tile_queue || queue4_ok) && (!ragged2x || ldw >= ceil_div64(K, 32) * 32) && K > 128 && K % 8 == 0 && N % 8 == 0 &&
        N >= 128 && max_rows >= 4096 && E <= 1024) {
      const int nt4 = (int)ceil_div64(N, BN4);
      const int64_t grid4 = (ceil_div64(max_rows, BM4) + E) * nt4;
      if (grid4 < 0x7fffffffLL) {
        const int gp = (int)std::min<int64_t>(grid4, device_cu_count());   // one persistent work-group per CU
        auto k4 = tile_queue ? (ragged2x ? grouped_gemm_nt4r_k<TO, true, true> : grouped_gemm_nt4r_k<TO, false, true>)
                             : (ragged2x ? grouped_gemm_nt4r_k<TO, true, false> : grouped_gemm_nt4r_k<TO, false, false>);
        const int lds4 = RING4 + 8 * 1024 + 64;   // ring + a bias area per wave + the queue's two hand-over words
        if (tile_queue && hipMemsetAsync(tile_queue, 0, NTQ_INTS * sizeof(int32_t), st) != hipSuccess) return APERTIS_ERR_LAUNCH;
        hipFuncSetAttribute((const void *)k4, hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
        // tile walk: groups of 8 m-tiles under panels of 2 n-tiles - the 32 work-groups of an XCD then work on 8 m-tiles x 4
        // n-tiles (2.9 MB of X + 1.4 MB of W: its L2).  Round 6 sweep of the footprint at 225 280 rows, N = 2816, K = 704
        // (avg of fc1 forward and fused fc2 data gradient, us; profiles/r6_probe_nt4r_walk.log): 8 x 4: 1228 | 4 x 8: 1250-1255 |
        // 16 x 2: 1270 | n-fastest (3 x 11): 1284 | 32 x 1: 1340
        int walk_g = nt4 > 4 ? 8 : 0, walk_nb = 2;
        hipLaunchKernelGGL(k4, dim3((unsigned)gp), dim3(NT4), lds4, st, (const bf16_t *)A, (const bf16_t *)W, bias, offsets,
                           (TO *)C, (TO *)pre_act, (const TO *)mul_pre, (int)N, (int)K, (int)ldw, (int)E, nt4, (int)grid4, act_flags,
                           drop_p, seed, walk_g, walk_nb, (int *)tile_queue);
        return apertis_check_launch();
      }
    }
    if (use2x && (!ragged2x || ldw >= ceil_div64(K, 32) * 32) && K >= 96 && K % 8 == 0 && N % 8 == 0 && N >= 64 && max_rows >= 4096 && E <= 1024) {
      const int nt3 = (int)ceil_div64(N, BN3);
      const int64_t grid3 = (ceil_div64(max_rows, BM3) + E) * nt3;
      if (grid3 < 0x7fffffffLL) {
        auto k3 = ragged2x ? grouped_gemm_nt2x_k<TO, true> : grouped_gemm_nt2x_k<TO, false>;
        hipFuncSetAttribute((const void *)k3, hipFuncAttributeMaxDynamicSharedMemorySize, RING3);
        // tile walk (tile_walk above): groups of 8 m-tiles under panels of 4 n-tiles when there are enough n-tiles - at
        // K = 704 the group's activation blocks (8 x 360 KB) and the weight panel (0.7 MB) together fit the XCD's 4 MiB L2.
        // Measured at 225 280 rows, N = 2816, K = 704 (tools/probes/gemm_probe.hip, PROBE_R3): n-fastest 1597-1653 us (fc1
        // forward) / 1255-1293 (fc2 data gradient); (8,4) 1379 / 1190; (32,4) 1374 / 1230; (12,11) 1400 / 1181; (16,8) 1471 / 1236
        int walk_g = nt3 > 8 ? 8 : 0, walk_nb = 4;
        int spread_fill = 0;
        hipLaunchKernelGGL(k3, dim3((unsigned)grid3), dim3(NT3), RING3, st, (const bf16_t *)A, (const bf16_t *)W, bias, offsets,
                           (TO *)C, (TO *)pre_act, (const TO *)mul_pre, (int)N, (int)K, (int)ldw, (int)E, nt3, act_flags, drop_p, seed,
                           walk_g, walk_nb, spread_fill);
        return apertis_check_launch();
      }
    }
    if (flagged) return APERTIS_ERR_UNSUPPORTED;
    if (kpad_ok && (N >= 512 || (E == 1 && N >= 128)) && max_rows >= 4096 && E <= 1024) {
      const int nt2 = (int)ceil_div64(N, BN2);
      const int64_t grid2 = (ceil_div64(max_rows, BM2) + E) * nt2;
      if (grid2 < 0x7fffffffLL) {
        const int ncu = device_cu_count();
        const int gp = (int)std::min<int64_t>(grid2, ncu);          // one persistent work-group per CU
        size_t ldsp = 4 * TILE2_BYTES + 4096 + 16;                   // ring + group offsets
        constexpr int solo = 4;   // K steps of a tile whose DMA the non-store waves issue alone (measured best of 2..6)
        if (tile_queue && hipMemsetAsync(tile_queue, 0, NTQ_INTS * sizeof(int32_t), st) != hipSuccess) return APERTIS_ERR_LAUNCH;
        auto kp = (K % 64 == 0 && ldw == K) ? grouped_gemm_nt256p_k<TO, false> : grouped_gemm_nt256p_k<TO, true>;
        hipFuncSetAttribute((const void *)kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp);
        hipLaunchKernelGGL(kp, dim3((unsigned)gp), dim3(NT2), ldsp, st, (const bf16_t *)A, (const bf16_t *)W, bias, offsets,
                           (TO *)C, (TO *)pre_act, (const TO *)mul_pre, (int)N, (int)K, (int)ldw, (int)E, nt2, (int)grid2, solo,
                           act, drop_p, seed, (int *)tile_queue);
        return apertis_check_launch();
      }
    }
  }
  if (flagged) return APERTIS_ERR_UNSUPPORTED;
  const size_t cstage = (size_t)BM * (BN * sizeof(TO) + 16);
  size_t lds = std::max<size_t>(4 * TILE_BYTES, cstage);
  auto kern = grouped_gemm_nt_k<T, TO>;
  if (lds > 64 * 1024) hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT), lds, st, (const T *)A, (const T *)W, bias, offsets, (TO *)C,
                     (TO *)pre_act, (const TO *)mul_pre, (int)N, (int)K, (int)ldw, (int)E, n_tiles, act, drop_p, seed);
  return apertis_check_launch();
}

}  // namespace

// 1 when apertis_grouped_gemm_nt takes the APERTIS_ACT_SAVE_GRAD / APERTIS_ACT_MUL_SAVED forms for this problem (the
// two-per-CU kernel's conditions in launch_nt: bf16, GELU, short K, wide N, enough rows); callers keep the pre-activation otherwise
extern "C" int apertis_grouped_gemm_nt_saves_grad(int64_t max_rows, int64_t N, int64_t K, int64_t ldw, int64_t E, int act,
                                                  int dtype, int dtype_out) {
  if (dtype != APERTIS_BF16 || dtype_out != APERTIS_BF16 || N <= 0 || K <= 0 || E <= 0 || act != APERTIS_ACT_GELU) return 0;
  if (ldw == 0) ldw = K;
  const bool ragged2x = K % 32 != 0;
  // ((max_rows + 256) * N < 2^32: the epilogue's mask hash takes 32-bit element indices)
  return K <= 1024 && N >= 512 && (!ragged2x || ldw >= ceil_div64(K, 32) * 32) && K >= 96 && K % 8 == 0 && N % 8 == 0 &&
         max_rows >= 4096 && E <= 1024 && (ceil_div64(max_rows, BM3) + E) * ceil_div64(N, BN3) < 0x7fffffffLL &&
         (max_rows + 256) * N < 0x100000000LL;
}

extern "C" int apertis_grouped_gemm_nt_q(const void *A, const void *W, const float *bias, const int32_t *offsets,
                                         void *C, void *pre_act, const void *act_bwd_pre, int64_t max_rows, int64_t N, int64_t K,
                                         int64_t ldw, int64_t E, int act, float drop_p, uint64_t seed, int dtype, int dtype_out,
                                         int32_t *tile_queue, void *stream) {
  if (!A || !W || !offsets || !C || max_rows < 0 || N <= 0 || K <= 0 || E <= 0 || (ldw != 0 && ldw < K)) return APERTIS_ERR_ARG;
  if (ldw == 0) ldw = K;
  if (drop_p < 0.f || drop_p >= 1.f || (act_bwd_pre && (pre_act || bias))) return APERTIS_ERR_ARG;
  if (max_rows == 0) return APERTIS_OK;
  if (max_rows > 0x7fffffffLL || N > 0x3fffffff || K > 0x3fffffff || ldw > 0x3fffffff || E > 4096) return APERTIS_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == APERTIS_BF16 && dtype_out == APERTIS_BF16) {
    if (K % 8 || N % 8) return APERTIS_ERR_UNSUPPORTED;
    return launch_nt<bf16_t, bf16_t>(A, W, bias, offsets, C, pre_act, act_bwd_pre, max_rows, N, K, ldw, E, act, drop_p, seed, tile_queue, st);
  }
  if (dtype == APERTIS_BF16 && dtype_out == APERTIS_F32) {
    if (K % 8 || N % 4) return APERTIS_ERR_UNSUPPORTED;
    return launch_nt<bf16_t, float>(A, W, bias, offsets, C, pre_act, act_bwd_pre, max_rows, N, K, ldw, E, act, drop_p, seed, tile_queue, st);
  }
  if (dtype == APERTIS_F32 && dtype_out == APERTIS_F32) {
    if (K % 4 || N % 4) return APERTIS_ERR_UNSUPPORTED;
    return launch_nt<float, float>(A, W, bias, offsets, C, pre_act, act_bwd_pre, max_rows, N, K, ldw, E, act, drop_p, seed, tile_queue, st);
  }
  return APERTIS_ERR_UNSUPPORTED;
}

extern "C" int apertis_grouped_gemm_nt(const void *A, const void *W, const float *bias, const int32_t *offsets,
                                       void *C, void *pre_act, const void *act_bwd_pre, int64_t max_rows, int64_t N, int64_t K,
                                       int64_t ldw, int64_t E, int act, float drop_p, uint64_t seed, int dtype, int dtype_out,
                                       void *stream) {
  return apertis_grouped_gemm_nt_q(A, W, bias, offsets, C, pre_act, act_bwd_pre, max_rows, N, K, ldw, E, act, drop_p, seed, dtype,
                                   dtype_out, nullptr, stream);
}

extern "C" int64_t apertis_grouped_gemm_tn_workspace_bytes(int64_t E, int n_problems) {
  if (E <= 0 || n_problems < 1 || n_problems > 2) return 0;
  const int ncu = device_cu_count();
  const int64_t groups = E * n_problems;
  if (groups > ncu) return 0;   // v3 does not apply; no workspace needed
  return (ncu / groups) * groups * (int64_t)std::max(TN3_SLOT, TN5_SLOT) * (int64_t)sizeof(float) + TN3_CTR_BYTES;
}

// Which v5 tile a ONE-group [M, N] weight gradient takes through apertis_grouped_gemm_tn (1 = 352 x 256, 0 = 256 x 352), or -1:
// the caller then cuts the rows into pseudo-groups for the 128 x 128 kernel and folds the partial sums itself (ops/gemm.py).
// Measured at 180 224 rows (tools/dense_wgrad_check.py; 128 x 128 kernel over pseudo-groups + fold -> this path): dW [704, 2816]
// 1105 -> 637 us (647 -> 1122 TF), [768, 768] 328 -> 279, [352, 704] 191 -> 166, [896, 224] 156 -> 160, [704, 176] 133 -> 141:
// every CU writes a whole 352 x 256 fp32 slice (93 MB of partial tiles + their fold per call, whatever the shape), so it pays
// from about a quarter of a million output elements on.  (Its first form lost everywhere below a dozen tiles - 234 us for
// dW [352, 704]: the fold of a single group's 85 slices per tile ran on 66 work-groups; it now takes one 1024-float chunk per
// work-group, and the work-groups that share a row slice are neighbours on one XCD.)
#ifndef TN_DENSE_MIN_FILL
#define TN_DENSE_MIN_FILL 60
#endif
#ifndef TN_DENSE_MIN_AREA
#define TN_DENSE_MIN_AREA 240000
#endif
extern "C" int apertis_grouped_gemm_tn_dense_variant(int64_t M, int64_t N) {
  if (M < 128 || N < 128 || M % 8 || N % 8 || M * N < TN_DENSE_MIN_AREA) return -1;
  auto area = [&](int64_t tm, int64_t tn) { return ceil_div64(M, tm) * ceil_div64(N, tn) * tm * tn; };
  const int64_t aw = area(352, 256), an = area(256, 352), best = std::min(aw, an);
  if (M * N * 100 < best * TN_DENSE_MIN_FILL) return -1;     // (more than 40 % of the MFMA work on zeros)
  return aw <= an ? 1 : 0;
}

extern "C" int apertis_grouped_gemm_tn_q(const void *A, const void *Bm, const int32_t *offsets, float *dW,
                                         float *dbias, int64_t max_rows, int64_t M, int64_t N, int64_t E, void *ws,
                                         int64_t ws_bytes, int dtype, int item_queue, void *stream) {
  if (!A || !Bm || !offsets || !dW || max_rows < 0 || M <= 0 || N <= 0 || E <= 0) return APERTIS_ERR_ARG;
  if (M > 0x3fffffff || N > 0x3fffffff || max_rows > 0x7fffffffLL) return APERTIS_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int m_tiles = (int)ceil_div64(M, BM), n_tiles = (int)ceil_div64(N, BN);
  const int64_t grid = (int64_t)E * m_tiles * n_tiles;
  if (grid > 0x7fffffffLL) return APERTIS_ERR_UNSUPPORTED;
  if (dtype == APERTIS_BF16) {
    if (M % 8 || N % 8 || !aligned16<bf16_t>(A, M) || !aligned16<bf16_t>(Bm, N)) return APERTIS_ERR_UNSUPPORTED;
    if (ws) {
      Tn3Problem q{(const bf16_t *)A, (const bf16_t *)Bm, dW, dbias, (int)M, (int)N, (int)ceil_div64(M, 256),
                   (int)ceil_div64(N, 256)};
      // One group (a dense layer's weight gradient: the reduction runs over ALL rows) with enough tiles: the 352-wide tiles of
      // v5, the rows of every tile split over the CUs and folded in slice order (apertis_grouped_gemm_tn_dense_variant).
      const int dv = apertis_grouped_gemm_tn_dense_variant(M, N);
      if (E == 1 && dv >= 0) {
        const int rc5 = launch_tn5(q, nullptr, E, max_rows, offsets, (float *)ws, ws_bytes, item_queue != 0, st, dv);
        if (rc5 != APERTIS_ERR_UNSUPPORTED) return rc5;
      }
      const int rc = launch_tn3(q, nullptr, E, max_rows, offsets, (float *)ws, ws_bytes, item_queue != 0, st);
      if (rc != APERTIS_ERR_UNSUPPORTED) return rc;
    }
    TnProblem q0{(const bf16_t *)A, (const bf16_t *)Bm, dW, dbias, (int)M, (int)N, m_tiles, n_tiles, (int)grid};
    TnProblem q1{nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0};
    return launch_tn2(q0, q1, offsets, st);
  } else if (dtype == APERTIS_F32) {
    if (M % 4 || N % 4 || !aligned16<float>(A, M) || !aligned16<float>(Bm, N)) return APERTIS_ERR_UNSUPPORTED;
    size_t lds = 2 * 32 * (BM * 4 + 16);
    hipLaunchKernelGGL(grouped_gemm_tn_k<float>, dim3((unsigned)grid), dim3(NT), lds, st, (const float *)A,
                       (const float *)Bm, offsets, dW, dbias, (int)M, (int)N, m_tiles, n_tiles);
  } else {
    return APERTIS_ERR_ARG;
  }
  return apertis_check_launch();
}

extern "C" int apertis_act_dropout_bwd(const void *dh, const void *pre_act, void *dpre, const int32_t *offsets,
                                       int64_t max_rows, int64_t N, int64_t E, int act, float drop_p, uint64_t seed,
                                       int dtype, void *stream) {
  if (!dh || !pre_act || !dpre || !offsets || max_rows < 0 || N <= 0 || E <= 0) return APERTIS_ERR_ARG;
  if (N % 4) return APERTIS_ERR_UNSUPPORTED;
  if (max_rows == 0) return APERTIS_OK;
  hipStream_t st = (hipStream_t)stream;
  int64_t nvec = max_rows * N / 4;
  int64_t nb = ceil_div64(nvec, 256);
  dim3 grid((unsigned)(nb < 8192 ? nb : 8192)), block(256);
  if (dtype == APERTIS_BF16)
    hipLaunchKernelGGL(act_dropout_bwd_k<bf16_t>, grid, block, 0, st, (const bf16_t *)dh, (const bf16_t *)pre_act,
                       (bf16_t *)dpre, offsets, max_rows, (int)N, (int)E, act, drop_p, seed);
  else if (dtype == APERTIS_F32)
    hipLaunchKernelGGL(act_dropout_bwd_k<float>, grid, block, 0, st, (const float *)dh, (const float *)pre_act,
                       (float *)dpre, offsets, max_rows, (int)N, (int)E, act, drop_p, seed);
  else
    return APERTIS_ERR_ARG;
  return apertis_check_launch();
}

extern "C" int apertis_grouped_gemm_tn(const void *A, const void *Bm, const int32_t *offsets, float *dW,
                                       float *dbias, int64_t max_rows, int64_t M, int64_t N, int64_t E, void *ws,
                                       int64_t ws_bytes, int dtype, void *stream) {
  return apertis_grouped_gemm_tn_q(A, Bm, offsets, dW, dbias, max_rows, M, N, E, ws, ws_bytes, dtype, 0, stream);
}

extern "C" int apertis_grouped_gemm_tn_pair_q(const void *A0, const void *B0, float *dW0, float *dbias0, int64_t M0,
                                              int64_t N0, const void *A1, const void *B1, float *dW1, float *dbias1,
                                              int64_t M1, int64_t N1, const int32_t *offsets, int64_t max_rows, int64_t E,
                                              void *ws, int64_t ws_bytes, int dtype, int item_queue, void *stream) {
  if (!A0 || !B0 || !dW0 || !A1 || !B1 || !dW1 || !offsets || max_rows < 0 || E <= 0) return APERTIS_ERR_ARG;
  if (dtype != APERTIS_BF16) {   // fp32 parity path: two ordinary launches
    int rc = apertis_grouped_gemm_tn(A0, B0, offsets, dW0, dbias0, max_rows, M0, N0, E, nullptr, 0, dtype, stream);
    return rc ? rc : apertis_grouped_gemm_tn(A1, B1, offsets, dW1, dbias1, max_rows, M1, N1, E, nullptr, 0, dtype, stream);
  }
  if (M0 <= 0 || N0 <= 0 || M1 <= 0 || N1 <= 0 || (M0 | N0 | M1 | N1) % 8) return APERTIS_ERR_UNSUPPORTED;
  if (!aligned16<bf16_t>(A0, M0) || !aligned16<bf16_t>(B0, N0) || !aligned16<bf16_t>(A1, M1) || !aligned16<bf16_t>(B1, N1))
    return APERTIS_ERR_UNSUPPORTED;
  if (ws && max_rows <= 0x7fffffffLL && std::max(std::max(M0, N0), std::max(M1, N1)) < 0x3fffffff) {
    Tn3Problem p0{(const bf16_t *)A0, (const bf16_t *)B0, dW0, dbias0, (int)M0, (int)N0, (int)ceil_div64(M0, 256),
                  (int)ceil_div64(N0, 256)};
    Tn3Problem p1{(const bf16_t *)A1, (const bf16_t *)B1, dW1, dbias1, (int)M1, (int)N1, (int)ceil_div64(M1, 256),
                  (int)ceil_div64(N1, 256)};
    bool v5 = true;
    if (v5) {   // the 704-wide family: 256 x 352 / 352 x 256 tiles
      const int rc5 = launch_tn5(p0, &p1, E, max_rows, offsets, (float *)ws, ws_bytes, item_queue != 0, (hipStream_t)stream);
      if (rc5 != APERTIS_ERR_UNSUPPORTED) return rc5;
    }
    const int rc = launch_tn3(p0, &p1, E, max_rows, offsets, (float *)ws, ws_bytes, item_queue != 0, (hipStream_t)stream);
    if (rc != APERTIS_ERR_UNSUPPORTED) return rc;
  }
  const int mt0 = (int)ceil_div64(M0, BM), nt0 = (int)ceil_div64(N0, BN), mt1 = (int)ceil_div64(M1, BM),
            nt1 = (int)ceil_div64(N1, BN);
  const int64_t g0 = E * mt0 * nt0, g1 = E * mt1 * nt1;
  if (g0 + g1 > 0x7fffffffLL) return APERTIS_ERR_UNSUPPORTED;
  TnProblem q0{(const bf16_t *)A0, (const bf16_t *)B0, dW0, dbias0, (int)M0, (int)N0, mt0, nt0, (int)g0};
  TnProblem q1{(const bf16_t *)A1, (const bf16_t *)B1, dW1, dbias1, (int)M1, (int)N1, mt1, nt1, (int)g1};
  return launch_tn2(q0, q1, offsets, (hipStream_t)stream);
}

extern "C" int apertis_grouped_gemm_tn_pair(const void *A0, const void *B0, float *dW0, float *dbias0, int64_t M0,
                                            int64_t N0, const void *A1, const void *B1, float *dW1, float *dbias1,
                                            int64_t M1, int64_t N1, const int32_t *offsets, int64_t max_rows, int64_t E,
                                            void *ws, int64_t ws_bytes, int dtype, void *stream) {
  return apertis_grouped_gemm_tn_pair_q(A0, B0, dW0, dbias0, M0, N0, A1, B1, dW1, dbias1, M1, N1, offsets, max_rows, E, ws,
                                        ws_bytes, dtype, 0, stream);
}
