// SCRATCH PROTOTYPE shell for scratch/gemm4w/gen.py: one workgroup (4 waves, one per SIMD) per 256 x 256 tile.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gemm4w_body.inc"

__global__ void __launch_bounds__(256, 1) gemm4w_kernel(const uint16_t* A, const uint16_t* W, uint16_t* C, int M, int N, int K) {
  extern __shared__ char smem[];
  const int tiles_m = M / 256, tiles_n = N / 256, nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {  // consecutive logical tiles (which share A / W panels) go to the same XCD
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int band = tiles_n <= 16 ? 1 : (tiles_m <= 16 ? tiles_m : 4);
  const int per_band = band * tiles_n;
  const int b0 = bid / per_band;
  const int rows_in_band = min(band, tiles_m - b0 * band);
  const int in_band = bid - b0 * per_band;
  const long m0 = (long)(b0 * band + in_band % rows_in_band) * 256, n0 = (long)(in_band / rows_in_band) * 256;
  const unsigned long long ap = (unsigned long long)(A + m0 * K), wp = (unsigned long long)(W + n0 * K);
  const unsigned long long cp = (unsigned long long)(C + m0 * N + n0);
  asm volatile(GEMM4W_BODY
               :
               : [tid] "v"(threadIdx.x), [a_lo] "s"((unsigned)ap), [a_hi] "s"((unsigned)(ap >> 32)), [w_lo] "s"((unsigned)wp),
                 [w_hi] "s"((unsigned)(wp >> 32)), [c_lo] "s"((unsigned)cp), [c_hi] "s"((unsigned)(cp >> 32)), [lda2] "s"(K * 2),
                 [ldc2] "s"(N * 2), [nloop] "s"(K / 128), [kmax] "s"((K / 64 - 1) * 128)
               : GEMM4W_CLOBBERS);
}

extern "C" int gemm4w(const uint16_t* A, const uint16_t* W, uint16_t* C, int M, int N, int K, void* stream) {
  if (M % 256 || N % 256 || K % 128 || K < 256) return -1;
  if ((long)256 * K * 2 >= (1L << 31)) return -2;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm4w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    attr = true;
  }
  gemm4w_kernel<<<(M / 256) * (N / 256), 256, 131072, (hipStream_t)stream>>>(A, W, C, M, N, K);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
