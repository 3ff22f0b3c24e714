// FLUX VAE decoder (diffusers AutoencoderKL, decoder half) -- the memory-bound kernels around the convolutions.
// The convolutions and the mid-block attention's projections / products run on the GEMM kernels (gemm.hip:
// mgx_conv3x3_nhwc, mgx_gemm); this file holds what sits between them.  Activations are NHWC bf16: "plain" = [H W][C]
// row-major, "padded" = [(H + 2)(W + 2)][C] with a zero border (the convolution's padding; only interiors are ever written).
//
// What it replaces: `vae.decode(latents)` of fastvideo/train_grpo_flux.py:279-289 (bf16 weights under bf16 autocast):
// GroupNorm(32 groups, eps 1e-6) runs in fp32 on the bf16 tensor and its fp32 result goes through SiLU in fp32 before the
// next convolution's input cast to bf16 -- one rounding, here at the store; nearest-neighbour upsampling is exact.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------- GroupNorm statistics
// stage 1: per-channel sum / sum of squares of a slab of rows; stage 2: groups, in double.
constexpr int GN_ROWS = 512;   // rows per stage-1 block

__global__ void __launch_bounds__(256) gn_partial_kernel(const bf16_raw* __restrict__ x, long ld, long M, int C,
                                                         float* __restrict__ part) {
  const int tpr = C / 8;                          // threads per row (8 channels = 16 bytes each)
  const int rpp = 256 / tpr;                      // rows per pass
  const int cx = threadIdx.x % tpr, ry = threadIdx.x / tpr;
  const long r0 = (long)blockIdx.x * GN_ROWS;
  const long r1 = r0 + GN_ROWS < M ? r0 + GN_ROWS : M;
  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
  if (ry < rpp)
    for (long r = r0 + ry; r < r1; r += rpp) {
      const uint4 u = *reinterpret_cast<const uint4*>(x + r * ld + cx * 8);
      const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = bf2f(w[e] & 0xffff), b = bf2f(w[e] >> 16);
        s[2 * e] += a; q[2 * e] += a * a;
        s[2 * e + 1] += b; q[2 * e + 1] += b * b;
      }
    }
  // rows of the block -> one value per channel (LDS: [ry][C] x 2)
  extern __shared__ float sh[];
  float* ss = sh;
  float* qq = sh + (size_t)rpp * C;
  if (ry < rpp) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ss[ry * C + cx * 8 + e] = s[e];
      qq[ry * C + cx * 8 + e] = q[e];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, b = 0.f;
    for (int r = 0; r < rpp; ++r) {
      a += ss[r * C + c];
      b += qq[r * C + c];
    }
    part[((long)blockIdx.x * 2 + 0) * C + c] = a;
    part[((long)blockIdx.x * 2 + 1) * C + c] = b;
  }
}

// stats[g] = (mean, rstd) of group g; one block per group
__global__ void __launch_bounds__(256) gn_finish_kernel(const float* __restrict__ part, float* __restrict__ stats, int nblk,
                                                        int C, int G, long M, float eps) {
  __shared__ double rs[4], rq[4];
  const int g = blockIdx.x, cpg = C / G;
  double s = 0.0, q = 0.0;
  for (long i = threadIdx.x; i < (long)nblk * cpg; i += 256) {
    const long b = i / cpg;
    const int c = g * cpg + (int)(i - b * cpg);
    s += (double)part[(b * 2 + 0) * C + c];
    q += (double)part[(b * 2 + 1) * C + c];
  }
  s = wave_sum_d(s);
  q = wave_sum_d(q);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { rs[w] = s; rq[w] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double n = (double)M * cpg;
    const double mean = (rs[0] + rs[1] + rs[2] + rs[3]) / n;
    double var = (rq[0] + rq[1] + rq[2] + rq[3]) / n - mean * mean;   // biased, like torch.nn.functional.group_norm
    if (var < 0.0) var = 0.0;
    stats[2 * g] = (float)mean;
    stats[2 * g + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }

// y = [silu]((x - mean) rstd w + b) -> bf16.  Output pixel (yy, xx) goes to out + (yy * out_row + xx) * out_px: a plain
// matrix (out_row = W * C, out_px = C) or the interior of a padded image (out_row = (W + 2) C, out pointing at pixel (1, 1)).
__global__ void __launch_bounds__(256) gn_apply_kernel(const bf16_raw* __restrict__ x, long ld, const float* __restrict__ stats,
                                                       const bf16_raw* __restrict__ gamma, const bf16_raw* __restrict__ beta,
                                                       bf16_raw* __restrict__ out, long out_row, long out_px, long M, int Wd,
                                                       int C, int G, int silu) {
  const int tpr = C / 8, cpg = C / G;
  const long total = M * tpr;
  for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
    const long r = id / tpr;
    const int c0 = (int)(id - r * tpr) * 8;
    const uint4 u = *reinterpret_cast<const uint4*>(x + r * ld + c0);
    const uint4 gw = *reinterpret_cast<const uint4*>(gamma + c0), gb = *reinterpret_cast<const uint4*>(beta + c0);
    const uint32_t xw[4] = {u.x, u.y, u.z, u.w}, ww[4] = {gw.x, gw.y, gw.z, gw.w}, bw[4] = {gb.x, gb.y, gb.z, gb.w};
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = c0 + 2 * e + h;
        const int g = c / cpg;
        const float xv = bf2f(h ? xw[e] >> 16 : xw[e] & 0xffff);
        const float wv = bf2f(h ? ww[e] >> 16 : ww[e] & 0xffff), bv = bf2f(h ? bw[e] >> 16 : bw[e] & 0xffff);
        float t = (xv - stats[2 * g]) * stats[2 * g + 1] * wv + bv;
        v[h] = silu ? silu_f(t) : t;
      }
      o[e] = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    }
    const long yy = r / Wd, xx = r - yy * Wd;
    *reinterpret_cast<uint4*>(out + yy * out_row + xx * out_px + c0) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// nearest-neighbour 2x: plain [H W][C] -> interior of a padded [(2H + 2)(2W + 2)][C] image (out points at pixel (1, 1))
__global__ void __launch_bounds__(256) upsample2x_pad_kernel(const bf16_raw* __restrict__ x, bf16_raw* __restrict__ out, int H,
                                                             int Wd, int C) {
  const int tpr = C / 8;
  const long total = (long)4 * H * Wd * tpr;
  const long orow = (long)(2 * Wd + 2) * C;
  for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
    const long p = id / tpr;
    const int c0 = (int)(id - p * tpr) * 8;
    const long oy = p / (2 * Wd), ox = p - oy * (2 * Wd);
    const uint4 u = *reinterpret_cast<const uint4*>(x + ((oy >> 1) * Wd + (ox >> 1)) * C + c0);
    *reinterpret_cast<uint4*>(out + oy * orow + ox * C + c0) = u;
  }
}

// latents [Cin][H][W] fp32 -> bf16 interior of a padded NHWC image with C channels (channels >= Cin stay zero)
__global__ void __launch_bounds__(256) nchw_to_pad_kernel(const float* __restrict__ z, bf16_raw* __restrict__ out, int Cin, int H,
                                                          int Wd, int C) {
  const long total = (long)H * Wd * Cin;
  const long orow = (long)(Wd + 2) * C;
  for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
    const int c = (int)(id % Cin);
    const long p = id / Cin;
    const long yy = p / Wd, xx = p - yy * Wd;
    out[yy * orow + xx * C + c] = f2bf(z[((long)c * H + yy) * Wd + xx]);
  }
}

// [H W][ld] bf16 (first Cout channels) -> image [Cout][H][W] bf16
__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const bf16_raw* __restrict__ x, long ld, bf16_raw* __restrict__ img,
                                                           int Cout, long HW) {
  const long total = HW * Cout;
  for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
    const int c = (int)(id / HW);
    const long p = id - (long)c * HW;
    img[id] = x[p * ld + c];
  }
}

// P[r, :] = bf16(softmax(scale * S[r, :])), S fp32 [M][ld]; one block per row, the row held in registers (n <= 256 * 64)
__global__ void __launch_bounds__(256) softmax_rows_kernel(const float* __restrict__ S, long lds_, bf16_raw* __restrict__ P,
                                                           long ldp, int n, float scale) {
  __shared__ float red[4];
  const float* row = S + (long)blockIdx.x * lds_;
  float v[64];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const int c = i * 256 + threadIdx.x;
    v[i] = c < n ? row[c] * scale : -INFINITY;
    mx = fmaxf(mx, v[i]);
  }
  mx = wave_max(mx);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    v[i] = __expf(v[i] - mx);                      // exp(-inf) = 0 beyond the row
    sum += v[i];
  }
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) red[w] = sum;
  __syncthreads();
  const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
  bf16_raw* prow = P + (long)blockIdx.x * ldp;
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const int c = i * 256 + threadIdx.x;
    if (c < n) prow[c] = f2bf(v[i] * inv);
  }
}

}  // namespace

extern "C" long mgx_group_norm_workspace(long M, int C, int G) { return (long)cdiv(M, GN_ROWS) * 2 * C + 2L * G; }

// y = [silu](GroupNorm(x)) for one image: x plain [M = H W][C] bf16 (leading dimension ld), G groups, affine gamma / beta [C]
// bf16; out: see gn_apply_kernel.  ws: fp32 scratch of mgx_group_norm_workspace(M, C, G) elements.
extern "C" int mgx_group_norm_nhwc(const uint16_t* x, long ld, const uint16_t* gamma, const uint16_t* beta, uint16_t* out,
                                   long out_row, long out_px, float* ws, int H, int Wd, int C, int G, float eps, int silu,
                                   void* stream) {
  MGX_REQUIRE(x && gamma && beta && out && ws && H > 0 && Wd > 0, "bad argument");
  MGX_REQUIRE(C % 8 == 0 && C <= 2048 && 256 % (C / 8) == 0 && G > 0 && C % G == 0, "channels must be 8 .. 2048, a power of two times 8");
  MGX_REQUIRE(ld % 8 == 0 && out_row % 8 == 0 && out_px % 8 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
                  ((uintptr_t)gamma % 16 == 0) && ((uintptr_t)beta % 16 == 0), "16-byte addressable rows");
  hipStream_t st = (hipStream_t)stream;
  const long M = (long)H * Wd;
  const int nblk = cdiv(M, GN_ROWS);
  float* part = ws;
  float* stats = ws + (long)nblk * 2 * C;
  const int rpp = 256 / (C / 8);
  gn_partial_kernel<<<nblk, 256, (size_t)2 * rpp * C * sizeof(float), st>>>(x, ld, M, C, part);
  gn_finish_kernel<<<G, 256, 0, st>>>(part, stats, nblk, C, G, M, eps);
  const long total = M * (C / 8);
  const int grid = (int)((total + 255) / 256 < 256L * 16 ? (total + 255) / 256 : 256L * 16);
  gn_apply_kernel<<<grid, 256, 0, st>>>(x, ld, stats, gamma, beta, out, out_row, out_px, M, Wd, C, G, silu);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int mgx_upsample2x_pad_nhwc(const uint16_t* x, uint16_t* out, int H, int Wd, int C, void* stream) {
  MGX_REQUIRE(x && out && H > 0 && Wd > 0 && C % 8 == 0, "bad argument");
  MGX_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0), "16-byte aligned images");
  const long total = (long)4 * H * Wd * (C / 8);
  const int grid = (int)((total + 255) / 256 < 256L * 16 ? (total + 255) / 256 : 256L * 16);
  upsample2x_pad_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, out, H, Wd, C);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int mgx_latents_to_pad_nhwc(const float* z, uint16_t* out, int Cin, int H, int Wd, int C, void* stream) {
  MGX_REQUIRE(z && out && Cin > 0 && Cin <= C && H > 0 && Wd > 0, "bad argument");
  const long total = (long)H * Wd * Cin;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  nchw_to_pad_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(z, out, Cin, H, Wd, C);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int mgx_nhwc_to_image(const uint16_t* x, long ld, uint16_t* img, int Cout, int H, int Wd, void* stream) {
  MGX_REQUIRE(x && img && Cout > 0 && ld >= Cout && H > 0 && Wd > 0, "bad argument");
  const long total = (long)H * Wd * Cout;
  const int grid = (int)((total + 255) / 256 < 256L * 16 ? (total + 255) / 256 : 256L * 16);
  nhwc_to_nchw_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, ld, img, Cout, (long)H * Wd);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int mgx_softmax_rows_f32(const float* S, long lds_, uint16_t* P, long ldp, int M, int n, float scale, void* stream) {
  MGX_REQUIRE(S && P && M > 0 && n > 0 && n <= 256 * 64 && lds_ >= n && ldp >= n, "row length must be <= 16384");
  softmax_rows_kernel<<<M, 256, 0, (hipStream_t)stream>>>(S, lds_, P, ldp, n, scale);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}
