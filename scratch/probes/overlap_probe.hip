// Do MFMA (waves 0-3) and softmax-like VALU work (waves 4-7) of DIFFERENT waves on one SIMD overlap for free?
// One workgroup of 8 waves per CU; wave w and w+4 share a SIMD.  mode bit 0: waves 0-3 run an MFMA loop, bit 1: waves
// 4-7 run an exp/fma/max loop.  Prints the three times.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void __launch_bounds__(512, 2) probe(float* out, int iters, int mode, int same) {
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool do_mfma = same ? (mode & 1) : ((mode & 1) && wid < 4);
  const bool do_valu = same ? (mode & 2) : ((mode & 2) && wid >= 4);
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  s16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + lane); b[i] = (short)(0x3f00 + i); }
  float x[32];
  for (int i = 0; i < 32; ++i) x[i] = 0.001f * (lane + i);
  float mx = 0.f, sum = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (do_mfma) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    }
    if (do_valu) {
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        mx = fmaxf(mx, x[i]);
        const float p = __builtin_amdgcn_exp2f(x[i] * 0.125f - mx);
        sum += p;
        x[i] = p * 0.5f + 0.25f;
      }
    }
    if (same == 2) __syncthreads();
  }
  float r = sum + mx;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) r += acc[j][i];
  for (int i = 0; i < 32; ++i) r += x[i];
  out[blockIdx.x * 512 + threadIdx.x] = r;
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int same = 0; same < 2; ++same)
    for (int mode = 1; mode <= 3; ++mode) {
      probe<<<256, 512>>>(out, 100, mode, same);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      probe<<<256, 512>>>(out, iters, mode, same);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("same_wave=%d mode=%d (%s%s): %.3f ms  (%.1f ns / iteration)\n", same, mode, mode & 1 ? "MFMA " : "", mode & 2 ? "VALU" : "",
             ms, ms * 1e6 / iters);
    }
  return 0;
}
