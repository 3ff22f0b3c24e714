// Scratch experiment (not product): sustained rate of back-to-back bf16 MFMAs of both shapes under the board's power cap.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
// a bf16 pair with random sign / mantissa and exponents in [2^-3, 2^0]
__device__ __forceinline__ uint32_t rnd_pair(uint32_t s, int zero) {
  if (zero) return 0;
  const uint32_t h = hash(s);
  const uint32_t lo = ((h & 0x80ffu) | (((124u + ((h >> 8) & 3u)) << 7) & 0x7f80u)) & 0xffffu;
  const uint32_t g = h >> 16;
  const uint32_t hi = ((g & 0x80ffu) | (((124u + ((g >> 8) & 3u)) << 7) & 0x7f80u)) & 0xffffu;
  return lo | (hi << 16);
}
__device__ __forceinline__ bf16x8 frag(uint32_t s, int zero) {
  u32x4 u = {rnd_pair(s, zero), rnd_pair(s + 1, zero), rnd_pair(s + 2, zero), rnd_pair(s + 3, zero)};
  return __builtin_bit_cast(bf16x8, u);
}

__global__ void __launch_bounds__(256) loop16(float* out, int iters, int zero) {
  const uint32_t s0 = (blockIdx.x * 256 + threadIdx.x) * 64;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = frag(s0 + 4 * i, zero); b[i] = frag(s0 + 16 + 4 * i, zero); }
  f32x4 acc[4][4] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
  }
  float r = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int k = 0; k < 4; ++k) r += acc[i][j][k];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

__global__ void __launch_bounds__(256) loop32(float* out, int iters, int zero) {
  const uint32_t s0 = (blockIdx.x * 256 + threadIdx.x) * 64;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = frag(s0 + 4 * i, zero); b[i] = frag(s0 + 16 + 4 * i, zero); }
  f32x16 acc[2][2] = {};
  for (int it = 0; it < iters; ++it) {
    // same FLOPs per iteration as loop16: 8 MFMAs of 32x32x16 (two K-steps over a 2x2 grid)
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(a[2 * k + i]), "v"(b[2 * k + j]));
  }
  float r = 0;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 16; ++k) r += acc[i][j][k];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

static float* g_out = nullptr;
extern "C" double run(int variant, int iters, int zero, int launches, int waves_per_simd) {
  if (!g_out) hipMalloc(&g_out, 4 * 256 * 512 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * waves_per_simd;
  hipEventRecord(e0, 0);
  for (int l = 0; l < launches; ++l) {
    if (variant == 16) loop16<<<grid, 256>>>(g_out, iters, zero);
    else loop32<<<grid, 256>>>(g_out, iters, zero);
  }
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)launches * grid * 4 * iters * 16 * 2.0 * 16 * 16 * 32;
  return flop / (ms * 1e-3) / 1e12;
}
