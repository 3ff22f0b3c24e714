// Small-M ("skinny") linears and tiny elementwise ops of the conditioning path (temb, AdaLN modulation vectors):
// M = batch (<= 16) rows against [N, K] weights.  These are weight-streaming, HBM-bound: each wave streams 16
// weight rows straight from global memory into MFMA fragments (no LDS round trip, 4 K-steps in flight) and the
// tiny activation matrix is read through L1.  Replaces the nn.Linear calls inside diffusers'
// CombinedTimestepGuidanceTextProjEmbeddings and AdaLayerNormZero*/AdaLayerNormContinuous `.linear`.
#include "../../include/mixgrpo_hip.h"
#include "common.h"

namespace {

// out[b, n] = bf16( sum_k x[b,k] W[n,k] + bias[n] ),  b < Bn <= 16
// Workgroup = 16 output features; its four waves split K (a wave walking all of K alone is latency-bound: 24 dependent
// round trips for K = 3072, 2.9 TB/s on the 113 MB modulation weights) and wave 0 adds the four partial tiles.
__global__ void __launch_bounds__(256) skinny_linear_kernel(const bf16_raw* __restrict__ x, long ldx,
                                                            const bf16_raw* __restrict__ W, long ldw,
                                                            const bf16_raw* __restrict__ bias,
                                                            bf16_raw* __restrict__ out, long ldo, int Bn, int N, int K) {
  __shared__ f32x4 red[3][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 16;
  const int fr = lane & 15, fq = lane >> 4;
  int nrow = n0 + fr;
  if (nrow >= N) nrow = N - 1;
  int brow = fr < Bn ? fr : Bn - 1;
  const int kchunk = ((K / 32 + 3) / 4) * 32;              // K-steps of 32 per wave, rounded up
  const int kbeg = min(w * kchunk, K), kend = min(kbeg + kchunk, K);
  const bf16_raw* wp = W + (long)nrow * ldw + fq * 8;
  const bf16_raw* xp = x + (long)brow * ldx + fq * 8;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int k = kbeg;
  for (; k + 128 <= kend; k += 128) {
    s16x8 a0 = *reinterpret_cast<const s16x8*>(wp + k), a1 = *reinterpret_cast<const s16x8*>(wp + k + 32);
    s16x8 a2 = *reinterpret_cast<const s16x8*>(wp + k + 64), a3 = *reinterpret_cast<const s16x8*>(wp + k + 96);
    s16x8 b0 = *reinterpret_cast<const s16x8*>(xp + k), b1 = *reinterpret_cast<const s16x8*>(xp + k + 32);
    s16x8 b2 = *reinterpret_cast<const s16x8*>(xp + k + 64), b3 = *reinterpret_cast<const s16x8*>(xp + k + 96);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b3, acc, 0, 0, 0);
  }
  for (; k < kend; k += 32) {
    s16x8 a0 = *reinterpret_cast<const s16x8*>(wp + k);
    s16x8 b0 = *reinterpret_cast<const s16x8*>(xp + k);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, acc, 0, 0, 0);
  }
  if (w > 0) red[w - 1][lane] = acc;
  __syncthreads();
  if (w > 0) return;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const f32x4 o = red[i][lane];
    acc[0] += o[0]; acc[1] += o[1]; acc[2] += o[2]; acc[3] += o[3];
  }
  // D[i = n local][j = b]: lane holds b = fr, n = n0 + 4*fq + r
  if (fr < Bn) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 4 * fq + r;
      if (n < N) {
        float v = acc[r];
        if (bias) v += bf2f(bias[n]);
        out[(long)fr * ldo + n] = f2bf(v);
      }
    }
  }
}

// dW[n, k] += sum_b dout[b, n] * x[b, k]  (fp32 accumulate into the gradient buffer); dbias[n] += sum_b dout[b, n]
__global__ void __launch_bounds__(256) skinny_wgrad_kernel(const bf16_raw* __restrict__ dout, long ldd,
                                                           const bf16_raw* __restrict__ x, long ldx,
                                                           float* __restrict__ dW, long ldw, float* __restrict__ dbias,
                                                           int Bn, int N, int K) {
  const int n = blockIdx.y;
  float dn[16];
  float sb = 0.f;
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    dn[b] = b < Bn ? bf2f(dout[(long)b * ldd + n]) : 0.f;
    sb += dn[b];
  }
  if (dbias && blockIdx.x == 0 && threadIdx.x == 0) dbias[n] += sb;
  for (int k = (blockIdx.x * 256 + threadIdx.x) * 4; k < K; k += gridDim.x * 256 * 4) {
    float4 g = *reinterpret_cast<float4*>(dW + (long)n * ldw + k);
    for (int b = 0; b < Bn; ++b) {
      const uint2 u = *reinterpret_cast<const uint2*>(x + (long)b * ldx + k);
      g.x += dn[b] * bf2f(u.x & 0xffff); g.y += dn[b] * bf2f(u.x >> 16);
      g.z += dn[b] * bf2f(u.y & 0xffff); g.w += dn[b] * bf2f(u.y >> 16);
    }
    *reinterpret_cast<float4*>(dW + (long)n * ldw + k) = g;
  }
}

// dx[b, k] (+)= bf16( sum_n dout[b, n] * W[n, k] ),  b < Bn <= 8: the input gradient of a skinny linear, straight from the
// row-major weight (a 128x128-tile GEMM on a transposed copy of W ran 24 workgroups and read W twice).
// Stage 1: workgroup (x, g) walks the weight rows n = g, g + G, ... of its 2048 columns (a thread = 8 columns, 16-byte loads);
// dout of those rows sits in LDS as fp32 [row][8].  Stage 2 sums the G partials in a fixed order.
constexpr int SD_G = 128;
__global__ void __launch_bounds__(256) skinny_dgrad_kernel(const bf16_raw* __restrict__ dout, long ldd,
                                                           const bf16_raw* __restrict__ W, long ldw, float* __restrict__ part,
                                                           int Bn, int N, int K) {
  extern __shared__ float dsh[];                           // [rows of this workgroup][8]
  const int g = blockIdx.y;
  const int nrows = (N - g + SD_G - 1) / SD_G;
  for (int id = threadIdx.x; id < nrows * 8; id += 256) {
    const int i = id >> 3, b = id & 7;
    dsh[id] = b < Bn ? bf2f(dout[(long)b * ldd + g + (long)i * SD_G]) : 0.f;
  }
  __syncthreads();
  const int c = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (c >= K) return;
  float acc[8][8];
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[b][e] = 0.f;
  const bf16_raw* wp = W + (long)g * ldw + c;
  const long step = (long)SD_G * ldw;
  int i = 0;
  for (; i + 4 <= nrows; i += 4) {
    uint4 u[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) u[q] = *reinterpret_cast<const uint4*>(wp + (i + q) * step);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float wv[8];
      wv[0] = bf2f(u[q].x & 0xffff); wv[1] = bf2f(u[q].x >> 16); wv[2] = bf2f(u[q].y & 0xffff); wv[3] = bf2f(u[q].y >> 16);
      wv[4] = bf2f(u[q].z & 0xffff); wv[5] = bf2f(u[q].z >> 16); wv[6] = bf2f(u[q].w & 0xffff); wv[7] = bf2f(u[q].w >> 16);
      const float4 d0 = *reinterpret_cast<const float4*>(dsh + (i + q) * 8), d1 = *reinterpret_cast<const float4*>(dsh + (i + q) * 8 + 4);
      const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
      for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[b][e] += dv[b] * wv[e];
    }
  }
  for (; i < nrows; ++i) {
    const uint4 u = *reinterpret_cast<const uint4*>(wp + i * step);
    float wv[8];
    wv[0] = bf2f(u.x & 0xffff); wv[1] = bf2f(u.x >> 16); wv[2] = bf2f(u.y & 0xffff); wv[3] = bf2f(u.y >> 16);
    wv[4] = bf2f(u.z & 0xffff); wv[5] = bf2f(u.z >> 16); wv[6] = bf2f(u.w & 0xffff); wv[7] = bf2f(u.w >> 16);
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[b][e] += dsh[i * 8 + b] * wv[e];
  }
  for (int b = 0; b < Bn; ++b) {
    float* pp = part + ((long)g * Bn + b) * K + c;
    *reinterpret_cast<float4*>(pp) = make_float4(acc[b][0], acc[b][1], acc[b][2], acc[b][3]);
    *reinterpret_cast<float4*>(pp + 4) = make_float4(acc[b][4], acc[b][5], acc[b][6], acc[b][7]);
  }
}

__global__ void __launch_bounds__(256) skinny_dgrad_finish_kernel(const float* __restrict__ part, bf16_raw* __restrict__ dx,
                                                                  long lddx, int Bn, int K, int accumulate) {
  const long id = (long)blockIdx.x * 256 + threadIdx.x;
  if (id >= (long)Bn * K) return;
  const int b = (int)(id / K), k = (int)(id - (long)b * K);
  float s = 0.f;
  for (int g = 0; g < SD_G; ++g) s += part[((long)g * Bn + b) * K + k];
  bf16_raw* o = dx + (long)b * lddx + k;
  const float r = rbf(s);                                  // the linear's backward output is a bf16 tensor ...
  *o = f2bf(accumulate ? bf2f(*o) + r : r);                // ... added to the running bf16 sum
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }

// op: 0 silu fwd (y = bf16(silu(a))) | 1 silu bwd (y = bf16(b * silu'(a))) | 2 add (y = bf16(a + b))
// 3 add-inplace-accumulate (y = bf16(y + a))
__global__ void ew_kernel(const bf16_raw* __restrict__ a, const bf16_raw* __restrict__ b, bf16_raw* __restrict__ y,
                          long n, int op) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float fa = bf2f(a[i]);
    float r;
    if (op == 0) r = silu_f(fa);
    else if (op == 1) {
      const float s = 1.0f / (1.0f + expf(-fa));
      r = bf2f(b[i]) * (s * (1.0f + fa * (1.0f - s)));
    } else if (op == 2) r = fa + bf2f(b[i]);
    else r = bf2f(y[i]) + fa;
    y[i] = f2bf(r);
  }
}

// Timesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0): out[b] = bf16([cos(t f_k) | sin(t f_k)]), f_k = 10000^(-k/128)
__global__ void sincos_embed_kernel(const float* __restrict__ t, bf16_raw* __restrict__ out, int Bn) {
  const int b = blockIdx.x, k = threadIdx.x;   // 128 threads
  if (b >= Bn) return;
  const float f = expf(-logf(10000.0f) * (float)k / 128.0f);
  const float ang = t[b] * f;
  out[(long)b * 256 + k] = f2bf(cosf(ang));
  out[(long)b * 256 + 128 + k] = f2bf(sinf(ang));
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_raw* __restrict__ y, long n) {
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4) {
    const float4 v = *reinterpret_cast<const float4*>(x + i);
    uint2 o;
    o.x = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
    o.y = (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16);
    *reinterpret_cast<uint2*>(y + i) = o;
  }
}

// y = scale * float(x): the all-reduced bf16 gradient bucket back into the flat fp32 gradient buffer (dist_utils.GradReducer)
__global__ void cast_bf16_f32_kernel(const bf16_raw* __restrict__ x, float* __restrict__ y, long n, float scale) {
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += (long)gridDim.x * blockDim.x * 8) {
    const uint4 u = *reinterpret_cast<const uint4*>(x + i);
    float4 a, b;
    a.x = bf2f(u.x & 0xffff) * scale; a.y = bf2f(u.x >> 16) * scale; a.z = bf2f(u.y & 0xffff) * scale; a.w = bf2f(u.y >> 16) * scale;
    b.x = bf2f(u.z & 0xffff) * scale; b.y = bf2f(u.z >> 16) * scale; b.z = bf2f(u.w & 0xffff) * scale; b.w = bf2f(u.w >> 16) * scale;
    *reinterpret_cast<float4*>(y + i) = a;
    *reinterpret_cast<float4*>(y + i + 4) = b;
  }
}

// gate-residual backward helpers on [rows, D] row-batched streams:
//   dgate[b, c] = sum_rows dout[m, c] * y[m, c]   (y = pre-gate branch output), dy[m, c] = bf16(gate[b, c] * dout[m, c])
// Block = 32 rows of one batch x 256 columns... each thread owns 1 column pair across the block's rows.
constexpr int GR_ROWS = 64;
__global__ void __launch_bounds__(256) gate_bwd_kernel(const bf16_raw* __restrict__ dout, long ldd, long d_rpb,
                                                       long d_bstride, const bf16_raw* __restrict__ y, long ldy,
                                                       const bf16_raw* __restrict__ gate, long gate_ld,
                                                       bf16_raw* __restrict__ dy, long lddy, float* __restrict__ part,
                                                       long rows_per_batch, int D, int blocks_per_batch) {
  const long b = blockIdx.y / blocks_per_batch;
  const long r0 = (long)(blockIdx.y % blocks_per_batch) * GR_ROWS;
  const int c = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (c >= D) return;
  const uint32_t gu = *reinterpret_cast<const uint32_t*>(gate + b * gate_ld + c);
  const float g0 = bf2f(gu & 0xffff), g1 = bf2f(gu >> 16);
  float a0 = 0.f, a1 = 0.f;
  const long rend = min(r0 + (long)GR_ROWS, rows_per_batch);
  for (long r = r0; r < rend; ++r) {
    const long m = b * rows_per_batch + r;
    const uint32_t du = *reinterpret_cast<const uint32_t*>(dout + b * d_bstride + r * ldd + c);
    const uint32_t yu = *reinterpret_cast<const uint32_t*>(y + m * ldy + c);
    const float d0 = bf2f(du & 0xffff), d1 = bf2f(du >> 16);
    a0 += d0 * bf2f(yu & 0xffff);
    a1 += d1 * bf2f(yu >> 16);
    *reinterpret_cast<uint32_t*>(dy + m * lddy + c) = (uint32_t)f2bf(g0 * d0) | ((uint32_t)f2bf(g1 * d1) << 16);
  }
  part[(long)blockIdx.y * D + c] = a0;
  part[(long)blockIdx.y * D + c + 1] = a1;
}

__global__ void __launch_bounds__(256) gate_bwd_finish_kernel(const float* __restrict__ part, bf16_raw* __restrict__ dgate,
                                                              long gate_ld, int D, int blocks_per_batch) {
  __shared__ float red[4][64];
  const int b = blockIdx.y;
  const int cx = threadIdx.x & 63, ky = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  float s = 0.f;
  if (c < D)
    for (int k = ky; k < blocks_per_batch; k += 4) s += part[((long)b * blocks_per_batch + k) * D + c];
  red[ky][cx] = s;
  __syncthreads();
  if (ky == 0 && c < D) dgate[(long)b * gate_ld + c] = f2bf((red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]));
}

}  // namespace

extern "C" int mgx_skinny_linear(const uint16_t* x, long ldx, const uint16_t* W, long ldw, const uint16_t* bias,
                                 uint16_t* out, long ldo, int Bn, int N, int K, void* stream) {
  MGX_REQUIRE(x && W && out, "null operand");
  MGX_REQUIRE(Bn >= 1 && Bn <= 16, "skinny linear handles 1..16 rows");
  MGX_REQUIRE(K % 32 == 0 && ldx % 8 == 0 && ldw % 8 == 0, "K must be a multiple of 32 with 16-byte aligned rows");
  skinny_linear_kernel<<<cdiv(N, 16), 256, 0, (hipStream_t)stream>>>(x, ldx, W, ldw, bias, out, ldo, Bn, N, K);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int mgx_skinny_wgrad(const uint16_t* dout, long ldd, const uint16_t* x, long ldx, float* dW, long ldw,
                                float* dbias, int Bn, int N, int K, void* stream) {
  MGX_REQUIRE(dout && x && dW, "null operand");
  MGX_REQUIRE(Bn >= 1 && Bn <= 16, "skinny wgrad handles 1..16 rows");
  MGX_REQUIRE(K % 4 == 0 && ldx % 4 == 0 && ldw % 4 == 0, "K must be a multiple of 4");
  int gx = cdiv(K, 1024);
  skinny_wgrad_kernel<<<dim3(gx, N), 256, 0, (hipStream_t)stream>>>(dout, ldd, x, ldx, dW, ldw, dbias, Bn, N, K);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" long mgx_skinny_dgrad_workspace(int Bn, int K) { return (long)SD_G * (Bn < 8 ? Bn : 8) * K; }

extern "C" int mgx_skinny_dgrad(const uint16_t* dout, long ldd, const uint16_t* W, long ldw, uint16_t* dx, long lddx,
                                float* ws, int Bn, int N, int K, int accumulate, void* stream) {
  MGX_REQUIRE(dout && W && dx && ws, "null operand");
  MGX_REQUIRE(Bn >= 1 && Bn <= 8, "skinny dgrad handles 1..8 rows per call");
  MGX_REQUIRE(N >= SD_G && K % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)W % 16 == 0), "weight rows must be 16-byte addressable");
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)cdiv(N, SD_G) * 8 * sizeof(float);
  MGX_REQUIRE(lds <= 65536, "too many weight rows");
  skinny_dgrad_kernel<<<dim3(cdiv(K, 2048), SD_G), 256, lds, st>>>(dout, ldd, W, ldw, ws, Bn, N, K);
  skinny_dgrad_finish_kernel<<<cdiv((long)Bn * K, 256), 256, 0, st>>>(ws, dx, lddx, Bn, K, accumulate);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int mgx_ew_bf16(const uint16_t* a, const uint16_t* b, uint16_t* y, long n, int op, void* stream) {
  MGX_REQUIRE(a && y && n > 0 && op >= 0 && op <= 3, "bad argument");
  MGX_REQUIRE(op == 0 || op == 3 || b, "binary op needs b");
  int nb = cdiv(n, 256);
  if (nb > 2048) nb = 2048;
  ew_kernel<<<nb, 256, 0, (hipStream_t)stream>>>(a, b, y, n, op);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int mgx_sincos_embed(const float* t, uint16_t* out, int Bn, void* stream) {
  MGX_REQUIRE(t && out && Bn > 0, "bad argument");
  sincos_embed_kernel<<<Bn, 128, 0, (hipStream_t)stream>>>(t, out, Bn);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int mgx_cast_f32_bf16(const float* x, uint16_t* y, long n, void* stream) {
  MGX_REQUIRE(x && y && n > 0 && n % 4 == 0, "bad argument");
  int nb = cdiv(n, 1024);
  if (nb > 4096) nb = 4096;
  cast_f32_bf16_kernel<<<nb, 256, 0, (hipStream_t)stream>>>(x, y, n);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int mgx_cast_bf16_f32(const uint16_t* x, float* y, long n, float scale, void* stream) {
  MGX_REQUIRE(x && y && n > 0 && n % 8 == 0, "bad argument");
  MGX_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0), "operands must be 16-byte aligned");
  int nb = cdiv(n, 2048);
  if (nb > 4096) nb = 4096;
  cast_bf16_f32_kernel<<<nb, 256, 0, (hipStream_t)stream>>>(x, y, n, scale);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" long mgx_gate_bwd_workspace(long batches, long rows_per_batch, int D) {
  return batches * ((rows_per_batch + GR_ROWS - 1) / GR_ROWS) * D;
}

extern "C" int mgx_gate_bwd(const uint16_t* dout, long ldd, long d_bstride, const uint16_t* y, long ldy,
                            const uint16_t* gate, long gate_ld, uint16_t* dy, long lddy, uint16_t* dgate, float* ws,
                            int batches, long rows_per_batch, int D, void* stream) {
  MGX_REQUIRE(dout && y && gate && dy && dgate && ws, "null operand");
  MGX_REQUIRE(D % 2 == 0 && batches > 0 && rows_per_batch > 0, "bad sizes");
  const int bpb = cdiv(rows_per_batch, GR_ROWS);
  hipStream_t st = (hipStream_t)stream;
  gate_bwd_kernel<<<dim3(cdiv(D, 512), batches * bpb), 256, 0, st>>>(dout, ldd, rows_per_batch, d_bstride, y, ldy, gate,
                                                                      gate_ld, dy, lddy, ws, rows_per_batch, D, bpb);
  gate_bwd_finish_kernel<<<dim3(cdiv(D, 64), batches), 256, 0, st>>>(ws, dgate, gate_ld, D, bpb);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}
