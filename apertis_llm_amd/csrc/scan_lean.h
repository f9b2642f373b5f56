// Pieces shared by the lean scan kernels (scan_gate.hip: three launches per direction; scan_lookback.hip: one launch per
// direction with a decoupled look-back): raw-buffer row access, bf16 packing, hardware softplus / silu, DPP quad moves, and
// the head of the look-back workspace.  Anonymous namespace: each translation unit gets its own copy.
#pragma once
#include "scan_common.h"

namespace {

typedef unsigned long long gran_t;
__device__ __forceinline__ void gran_store(gran_t *p, uint32_t epoch, float v) {
  __hip_atomic_store(p, ((gran_t)epoch << 32) | (gran_t)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ gran_t gran_load(const gran_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// head of the look-back workspace (64 bytes), then the granules [B*ncs][nchunks + nsuper][CW*64 lanes][2]
struct GateWsHead { unsigned ctr[2]; int err; int pad[13]; };

// silu and its derivative on the hardware exp2 / rcp (1 ulp each: ~3e-7 relative, far inside the 1e-4 parity bar); the
// stand-alone gate kernel's expf + IEEE division cost ~30 VALU instructions per element
__device__ __forceinline__ float sigmoid_g(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-x * LOG2E_F)); }
__device__ __forceinline__ float silu_g(float x) { return x * sigmoid_g(x); }
__device__ __forceinline__ void silu_both(float x, float &f, float &df) {
  const float s = sigmoid_g(x);
  f = x * s;
  df = s * (1.f + x * (1.f - s));
}

// Rows come through raw buffer descriptors over each tensor (slice): a lane's offset is (its item's first token, its four
// channels) as ONE 32-bit register per tensor and the token's row term rides in the scalar offset - no per-token address
// arithmetic at all (64-bit pointers cost the first form of these kernels 214 VGPRs).  The launcher takes this path only for
// tensors below 4 GiB.
typedef unsigned lean_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t lean_rsrc(const void *p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint2 lean_ld8(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
  const lean_u2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, (int)soff, 2);   // (non-temporal: read once)
  return make_uint2(v[0], v[1]);
}
__device__ __forceinline__ float lean_ld4(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)soff, 0));
}
// softplus on the hardware exp2 / log2: max(x, 0) + log1p(exp(-|x|)), the log1p by the u = 1 + e trick (log(u) * e / (u - 1):
// exact where u - 1 == e, ~1e-7 relative elsewhere) - the library log1pf(expf(x)) is ~50 instructions per (token, head)
__device__ __forceinline__ float softplus_fast(float x) {
  const float e = __builtin_amdgcn_exp2f(-fabsf(x) * LOG2E_F);
  const float u = 1.f + e;
  float r = __builtin_amdgcn_logf(u) * 0.6931471805599453f;
  const float um1 = u - 1.f;
  r = um1 > 0.f ? r * (e * __builtin_amdgcn_rcpf(um1)) : e;
  return fmaxf(x, 0.f) + r;
}

__device__ __forceinline__ void unpack4(uint2 v, float (&f)[4]) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
}
__device__ __forceinline__ uint32_t lean_pack2(float a, float b) {
  typedef float f2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
  const f2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2_t));
}
struct LeanT { const void *p; uint32_t rs, bytes; };   // tensor (slice): base, row stride in elements, extent in bytes

// delta of FOUR consecutive tokens per wave-instruction: the four lanes of a head (N = 16: a quad) would each run the same
// softplus for the same token - instead lane i of the quad loads and transforms token t0 + i, and the per-token value is a
// quad broadcast (one DPP move).
template <int I> __device__ __forceinline__ float quad_bc(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), I | (I << 2) | (I << 4) | (I << 6), 0xf, 0xf, true));
}
__device__ __forceinline__ float quad_bc(float v, int i) {
  return i == 0 ? quad_bc<0>(v) : i == 1 ? quad_bc<1>(v) : i == 2 ? quad_bc<2>(v) : quad_bc<3>(v);
}

__device__ __forceinline__ float quad_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  return v;
}

}  // namespace
