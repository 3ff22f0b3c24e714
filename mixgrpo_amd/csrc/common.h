// Shared device/host helpers for the MixGRPO MI355X (gfx950) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>

// Diagnostic builds (-DMGX_DIAGNOSTIC_BUILD: mgx_version() < 0, refused by mixgrpo_amd/_lib.py) are written to scratch/ only
// (mixgrpo_amd/build.py --diagnostic).  The product sources carry no timing-only switches: in-kernel stamps and timing-only
// variants of the generated kernels are options of their generators (csrc/gen/*.py --diag) built by scratch harnesses.

typedef __bf16 bf16_t;
typedef uint16_t bf16_raw;  // storage view used on the C ABI

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

#define MGX_OK 0
#define MGX_ERR_ARG (-1)
#define MGX_ERR_LAUNCH (-2)
#define MGX_ERR_UNSUPPORTED (-3)

extern "C" void mgx_set_error(const char* msg);

#define MGX_REQUIRE(cond, msg)                                                    \
  do {                                                                            \
    if (!(cond)) {                                                                \
      char _b[512];                                                               \
      snprintf(_b, sizeof(_b), "%s:%d: %s (%s)", __FILE__, __LINE__, msg, #cond); \
      mgx_set_error(_b);                                                          \
      return MGX_ERR_ARG;                                                         \
    }                                                                             \
  } while (0)

#define MGX_CHECK_LAUNCH()                                                      \
  do {                                                                          \
    hipError_t _e = hipGetLastError();                                          \
    if (_e != hipSuccess) {                                                     \
      char _b[512];                                                             \
      snprintf(_b, sizeof(_b), "%s:%d: %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
      mgx_set_error(_b);                                                        \
      return MGX_ERR_LAUNCH;                                                    \
    }                                                                           \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_raw u) { return __builtin_bit_cast(float, (uint32_t)u << 16); }
__device__ __forceinline__ bf16_raw f2bf(float f) { return __builtin_bit_cast(bf16_raw, (bf16_t)f); }
// round a float to the nearest bf16 and back (the reference's bf16 intermediate tensors)
__device__ __forceinline__ float rbf(float f) { return (float)(bf16_t)f; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
