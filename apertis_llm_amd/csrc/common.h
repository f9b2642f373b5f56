// Shared device/host helpers for the gfx950 kernels.  gfx950 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/apertis_hip.h"

#define APERTIS_WAVE 64

typedef __bf16 bf16_t;

template <typename T> struct dtype_of;
template <> struct dtype_of<float> { static constexpr int value = APERTIS_F32; };
template <> struct dtype_of<bf16_t> { static constexpr int value = APERTIS_BF16; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
// plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving) on gfx950
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

static inline int apertis_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? APERTIS_OK : APERTIS_ERR_LAUNCH;
}

__host__ __device__ static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// largest power-of-two byte width (<=16) that divides every value in the list
static inline int common_align(std::initializer_list<uint64_t> vals) {
  uint64_t o = 0;
  for (uint64_t v : vals) o |= v;
  int a = 16;
  while (a > 1 && (o & (uint64_t)(a - 1))) a >>= 1;
  return a;
}

// counter-based keep mask: 16 random bits per element from a 32-bit avalanche of
// (element pair index, seed); the backward regenerates it from the same (seed,row,col)
__device__ __forceinline__ bool drop_keep(uint64_t seed, int64_t row, int64_t col, int64_t ncols, uint32_t thresh16) {
  uint64_t lin = (uint64_t)row * (uint64_t)ncols + (uint64_t)col;
  uint32_t h = (uint32_t)(lin >> 1) ^ (uint32_t)seed;
  h += (uint32_t)(lin >> 33) * 0x9E3779B9u + (uint32_t)(seed >> 32);
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  uint32_t r16 = (lin & 1) ? (h >> 16) : (h & 0xffffu);
  return r16 >= thresh16;
}

// the same mask for 4 consecutive elements starting at a linear index that is a multiple of 4:
// two hashes instead of four
__device__ __forceinline__ uint32_t drop_hash_pair(uint64_t seed, uint64_t pair) {
  uint32_t h = (uint32_t)pair ^ (uint32_t)seed;
  h += (uint32_t)(seed >> 32);
  // the high word is zero below 2^33 elements: a wave-uniform branch (hipcc if-converts a per-lane one and keeps
  // the quarter-rate multiply) skips its term
  uint32_t hi = (uint32_t)(pair >> 32);
  if (__builtin_amdgcn_ballot_w64(hi != 0u)) {
    asm volatile("" : "+v"(hi));
    h += hi * 0x9E3779B9u;
  }
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ void drop_keep4(uint64_t seed, uint64_t lin0, uint32_t thresh16, bool (&keep)[4]) {
  const uint32_t h0 = drop_hash_pair(seed, lin0 >> 1), h1 = drop_hash_pair(seed, (lin0 >> 1) + 1);
  keep[0] = (h0 & 0xffffu) >= thresh16; keep[1] = (h0 >> 16) >= thresh16;
  keep[2] = (h1 & 0xffffu) >= thresh16; keep[3] = (h1 >> 16) >= thresh16;
}

#define LOG2E_F 1.4426950408889634f
