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

#define LOG2E_F 1.4426950408889634f
