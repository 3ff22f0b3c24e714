// Scratch experiment: does the lane -> address map of the GEMM epilogue's 16-byte stores limit the CU's store path?
// Every wave stores blocks of 128 rows x 128 bytes (its 64 output features) into a row-major matrix with leading dimension ld:
//   pattern 0 (shipped epilogue): instruction (j, h): lane (fr = l & 15, fq = l >> 4) -> row 16 j + fr, bytes 64 h + 16 fq
//   pattern 1 (full lines):       instruction i:      lane l -> row 8 i + (l >> 3), bytes 16 (l & 7)
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void __launch_bounds__(512) store_kernel(char* out, long ld_bytes, int rows_total, int col_blocks, int pattern, int reps) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
  // workgroup tile = 256 rows x 512 bytes (8 waves: 2 row halves x 4 column blocks), walking tiles like the persistent GEMM
  const int tiles_n = col_blocks / 4;
  const int tiles_m = rows_total / 256;
  for (int rep = 0; rep < reps; ++rep)
    for (int t = blockIdx.x; t < tiles_m * tiles_n; t += gridDim.x) {
      const long m0 = (long)(t / tiles_n) * 256 + (wid >> 2) * 128;
      const long c0 = ((long)(t % tiles_n) * 4 + (wid & 3)) * 128;
      char* base = out + m0 * ld_bytes + c0;
      if (pattern == 0) {
        const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          *reinterpret_cast<uint4*>(base + (long)(16 * j + fr) * ld_bytes + 16 * fq) = v;
          *reinterpret_cast<uint4*>(base + (long)(16 * j + fr) * ld_bytes + 64 + 16 * fq) = v;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) *reinterpret_cast<uint4*>(base + (long)(8 * i + (lane >> 3)) * ld_bytes + 16 * (lane & 7)) = v;
      }
      v.x += 1;
    }
}
static char* g_buf = nullptr;
extern "C" double run(int pattern, int M, int N, int reps) {
  const size_t bytes = (size_t)M * N * 2;
  if (!g_buf) hipMalloc(&g_buf, (size_t)36864 * 12288 * 2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  store_kernel<<<256, 512>>>(g_buf, (long)N * 2, M, N * 2 / 128, pattern, 1);
  hipEventRecord(e0, 0);
  store_kernel<<<256, 512>>>(g_buf, (long)N * 2, M, N * 2 / 128, pattern, reps);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return (double)bytes * reps / (ms * 1e-3) / 1e12;   // TB/s
}
