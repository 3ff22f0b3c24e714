// Shared device/host helpers for the MixGRPO MI355X (gfx950) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>

// Timing-only diagnostic switches (-DMGX_TIMING_ONLY_*, -DMGX_GEMM_COMPILER_WAITS) compile kernels whose RESULTS ARE WRONG
// (they price one part of a kernel by removing it).  They exist only together with -DMGX_DIAGNOSTIC_BUILD, which makes
// mgx_version() negative: mixgrpo_amd/_lib.py refuses such a library, and mixgrpo_amd/build.py writes it to scratch/ only.
#if (defined(MGX_TIMING_ONLY_NO_EPI_STORES) || defined(MGX_TIMING_ONLY_NO_EPILOGUE) || defined(MGX_TIMING_ONLY_NO_DMA) ||   \
     defined(MGX_TIMING_ONLY_NO_FRAG_READS) || defined(MGX_TIMING_ONLY_MFMA32) || defined(MGX_TR_MAP16X4) || defined(MGX_TIMING_ONLY_NO_KTILE_SYNC) || \
     defined(MGX_TIMING_ONLY_NO_VMCNT) || defined(MGX_TIMING_ONLY_PP_NODMA) || defined(MGX_TIMING_ONLY_PP_NOREADS) || defined(MGX_TIMING_ONLY_NO_BARRIER) || defined(MGX_GEMM_COMPILER_WAITS) ||        \
     defined(MGX_GEMM_SETPRIO) || defined(MGX_GEMM_WSTAG) || defined(MGX_DIAG_DKV_STAMPS) || defined(MGX_DIAG_PP_STAMPS) || defined(MGX_DIAG_PP_CLOCK) || defined(MGX_DIAG_FWD_STAMPS) || defined(MGX_GEMM_A_AUX) || defined(MGX_GEMM_W_AUX) || defined(MGX_EPI_LAYOUT16)) &&       \
    !defined(MGX_DIAGNOSTIC_BUILD)
#error "MGX_TIMING_ONLY_* / MGX_GEMM_COMPILER_WAITS need -DMGX_DIAGNOSTIC_BUILD (python -m mixgrpo_amd.build --diagnostic ...)"
#endif

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
