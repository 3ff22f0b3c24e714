// HBM-bound normalisation kernels of the FLUX MMDiT (d = 3072 rows, head_dim 128), forward and backward:
//   * AdaLN modulate: y = bf16( LayerNorm(x; eps 1e-6, no affine) * bf16(1 + scale[b]) + shift[b] )
//   * QK RMSNorm (learned weight[128], eps 1e-6) + RoPE (interleaved pairs) + head split, and V transposition
// All loads/stores are 16-byte vectors; one wave per row (LayerNorm) or per (token, head) (RMSNorm).
// Replaces diffusers' AdaLayerNormZero / AdaLayerNormZeroSingle / AdaLayerNormContinuous normalisation parts,
// RMSNorm (norm_q/norm_k/norm_added_q/norm_added_k) and apply_rotary_emb inside FluxAttnProcessor2_0.
#include "../../include/mixgrpo_hip.h"
#include "common.h"

namespace {

struct RowMap {
  long ld, rpb, bstride;
};
__device__ __forceinline__ long row_off(const RowMap& r, long m) {
  const long b = m / r.rpb;
  return b * r.bstride + (m - b * r.rpb) * r.ld;
}

constexpr int LN_MAXV = 8;   // up to 8 x 8 elements per lane = 4096 columns

__device__ __forceinline__ void unpack8(const uint4& u, float* f) {
  f[0] = bf2f(u.x & 0xffff); f[1] = bf2f(u.x >> 16); f[2] = bf2f(u.y & 0xffff); f[3] = bf2f(u.y >> 16);
  f[4] = bf2f(u.z & 0xffff); f[5] = bf2f(u.z >> 16); f[6] = bf2f(u.w & 0xffff); f[7] = bf2f(u.w >> 16);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 u;
  u.x = (uint32_t)f2bf(f[0]) | ((uint32_t)f2bf(f[1]) << 16);
  u.y = (uint32_t)f2bf(f[2]) | ((uint32_t)f2bf(f[3]) << 16);
  u.z = (uint32_t)f2bf(f[4]) | ((uint32_t)f2bf(f[5]) << 16);
  u.w = (uint32_t)f2bf(f[6]) | ((uint32_t)f2bf(f[7]) << 16);
  return u;
}

// ------------------------------------------------------------------------------------- LN-modulate forward
// one wave per row; D = 8 * 64 * NV columns handled as NV vectors of 8 per lane (D % 512 == 0)
template <int NV>
__global__ void __launch_bounds__(256) ln_mod_fwd_kernel(const bf16_raw* __restrict__ x, RowMap xm,
                                                         const bf16_raw* __restrict__ shift,
                                                         const bf16_raw* __restrict__ scale, long mod_ld,
                                                         bf16_raw* __restrict__ y, long ldy, float* __restrict__ stats,
                                                         long M) {
  const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const int lane = threadIdx.x & 63;
  const long b = m / xm.rpb;
  const bf16_raw* xr = x + row_off(xm, m);
  float v[NV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    unpack8(*reinterpret_cast<const uint4*>(xr + (i * 64 + lane) * 8), v[i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[i][j];
  }
  const float D = (float)(NV * 512);
  const float mean = wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = v[i][j] - mean;
      q += d * d;
    }
  const float rstd = rsqrtf(wave_sum(q) / D + 1e-6f);
  if (stats && lane == 0) {
    stats[2 * m] = mean;
    stats[2 * m + 1] = rstd;
  }
  const bf16_raw* sh = shift + b * mod_ld;
  const bf16_raw* sc = scale + b * mod_ld;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 8;
    float fs[8], fc[8], o[8];
    unpack8(*reinterpret_cast<const uint4*>(sh + c), fs);
    unpack8(*reinterpret_cast<const uint4*>(sc + c), fc);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mean) * rstd * rbf(1.0f + fc[j]) + fs[j];
    *reinterpret_cast<uint4*>(y + m * ldy + c) = pack8(o);
  }
}

// ------------------------------------------------------------------------------------- LN-modulate backward
// dy [M, D] -> dx accumulated into the residual-stream gradient (row-batched, bf16), and per-batch column sums
// dshift[b, :] += sum_rows dy, dscale[b, :] += sum_rows dy * xhat (two-stage, deterministic).
// Each block owns ROWS_PER_BLOCK consecutive rows of ONE batch; each wave walks rows w, w+4, ...
constexpr int LN_BWD_ROWS = 32;
template <int NV>
__global__ void __launch_bounds__(256) ln_mod_bwd_kernel(const bf16_raw* __restrict__ dy, long lddy,
                                                         const bf16_raw* __restrict__ x, RowMap xm,
                                                         const bf16_raw* __restrict__ scale, long mod_ld,
                                                         bf16_raw* __restrict__ dx, RowMap dxm, int accumulate,
                                                         float* __restrict__ part, long M, int blocks_per_batch) {
  extern __shared__ float red[];   // [2][D]
  const int D = NV * 512;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long b = blockIdx.x / blocks_per_batch;
  const long r0 = (long)(blockIdx.x % blocks_per_batch) * LN_BWD_ROWS;
  const bf16_raw* sc = scale + b * mod_ld;
  float a_sh[NV][8], a_sc[NV][8];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) a_sh[i][j] = a_sc[i][j] = 0.f;

  // Row pipeline: the kernel holds 96 fp32 column accumulators + the row in fp32 (346 registers: ONE wave per SIMD, four waves
  // per CU), so memory parallelism has to come from inside the wave: the NEXT row's operands -- x, dy and the running gradient
  // the result is added to (it used to be read after the three reductions: a second round trip per row) -- are requested,
  // packed, before the current row's arithmetic starts.  (Keeping the row packed and forcing two waves per SIMD spills 63-96
  // registers: tried, not kept.)  Same operations on the same values as before: bit-identical results.
  auto row_ok = [&](int rr) { return rr < LN_BWD_ROWS && r0 + rr < xm.rpb && b * xm.rpb + r0 + rr < M; };
  uint4 nx[NV], ng[NV], nold[NV];
  auto fetch = [&](int rr) {
    const long m = b * xm.rpb + r0 + rr;
    const bf16_raw* xr = x + row_off(xm, m);
    const bf16_raw* dxr = dx + row_off(dxm, m);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      nx[i] = *reinterpret_cast<const uint4*>(xr + (i * 64 + lane) * 8);
      ng[i] = *reinterpret_cast<const uint4*>(dy + m * lddy + (i * 64 + lane) * 8);
      nold[i] = accumulate ? *reinterpret_cast<const uint4*>(dxr + (i * 64 + lane) * 8) : make_uint4(0, 0, 0, 0);
    }
  };
  if (row_ok(w)) fetch(w);
  for (int rr = w; row_ok(rr); rr += 4) {
    const long m = b * xm.rpb + r0 + rr;
    bf16_raw* dxr = dx + row_off(dxm, m);
    float v[NV][8], g[NV][8];
    uint4 oldp[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      unpack8(nx[i], v[i]);
      unpack8(ng[i], g[i]);
      oldp[i] = nold[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[i][j];
    }
    if (row_ok(rr + 4)) fetch(rr + 4);               // wave-uniform
    const float Df = (float)D;
    const float mean = wave_sum(s) / Df;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = v[i][j] - mean;
        q += d * d;
      }
    const float rstd = rsqrtf(wave_sum(q) / Df + 1e-6f);
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float fc[8];
      unpack8(*reinterpret_cast<const uint4*>(sc + (i * 64 + lane) * 8), fc);   // L1/L2 resident
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (v[i][j] - mean) * rstd;
        const float gx = g[i][j] * rbf(1.0f + fc[j]);   // d/dxhat
        a_sh[i][j] += g[i][j];
        a_sc[i][j] += g[i][j] * xh;
        c1 += gx;
        c2 += gx * xh;
        v[i][j] = xh;
        g[i][j] = gx;
      }
    }
    c1 = wave_sum(c1) / Df;
    c2 = wave_sum(c2) / Df;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = rstd * (g[i][j] - c1 - v[i][j] * c2);
      uint4* p = reinterpret_cast<uint4*>(dxr + (i * 64 + lane) * 8);
      if (accumulate) {
        float old[8];
        unpack8(oldp[i], old);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += old[j];
      }
      *p = pack8(o);
    }
  }
  // combine the 4 waves' column sums wave after wave in one [2][D] LDS image, emit one partial row per block
  for (int ww = 0; ww < 4; ++ww) {
    if (w == ww) {
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = (i * 64 + lane) * 8 + j;
          if (ww == 0) { red[c] = a_sh[i][j]; red[D + c] = a_sc[i][j]; }
          else { red[c] += a_sh[i][j]; red[D + c] += a_sc[i][j]; }
        }
    }
    __syncthreads();
  }
  for (int c = threadIdx.x; c < 2 * D; c += 256) part[(long)blockIdx.x * 2 * D + c] = red[c];
}

// dshift[b, c] (+)= sum over the batch's blocks; gradients are bf16 [B, mod_ld] chunks of the modulation vector.
// Block = 64 columns x 4 lanes over the blocks (a thread per column walking all partials alone is latency-bound).
__global__ void __launch_bounds__(256) ln_mod_bwd_finish_kernel(const float* __restrict__ part, bf16_raw* __restrict__ dshift,
                                                                bf16_raw* __restrict__ dscale, long mod_ld, int D,
                                                                int blocks_per_batch) {
  __shared__ float red[2][4][64];
  const int b = blockIdx.y;
  const int cx = threadIdx.x & 63, ky = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  float s0 = 0.f, s1 = 0.f;
  if (c < D)
    for (int k = ky; k < blocks_per_batch; k += 4) {
      const long blk = (long)b * blocks_per_batch + k;
      s0 += part[(blk * 2 + 0) * D + c];
      s1 += part[(blk * 2 + 1) * D + c];
    }
  red[0][ky][cx] = s0;
  red[1][ky][cx] = s1;
  __syncthreads();
  if (ky == 0 && c < D) {
    dshift[(long)b * mod_ld + c] = f2bf((red[0][0][cx] + red[0][1][cx]) + (red[0][2][cx] + red[0][3][cx]));
    dscale[(long)b * mod_ld + c] = f2bf((red[1][0][cx] + red[1][1][cx]) + (red[1][2][cx] + red[1][3][cx]));
  }
}

// ------------------------------------------------------------------------------------- QK norm + RoPE + split
// qkv [rows, 3*H*128] (q | k | v per token, plain matrix of one stream) -> Q, K [B, H, S, 128], Vt [B, H, 128, Sp]
// Block = 64 tokens x 1 head; 4 waves: each wave 16 tokens; lane = one rotation pair (2 elements).
struct QkArgs {
  const bf16_raw* qkv;
  long ld;            // 3*H*128
  const float* wq;    // [128] fp32 RMSNorm weights
  const float* wk;
  const float* cos;   // [S, 128] fp32
  const float* sin;
  bf16_raw* Q;
  bf16_raw* K;
  bf16_raw* Vt;
  bf16_raw* V;    // optional extras for the backward pass: V row-major [B,H,S,128], Qt/Kt [B,H,128,Sp]
  bf16_raw* Qt;
  bf16_raw* Kt;
  int H, S, Sp;
  int rows_per_batch;  // tokens of this stream per sample
  int s0;              // position of the stream's first token in the joint sequence
  float q_scale;       // Q (and Qt) leave as bf16(q * q_scale): 1, or softmax scale * log2(e) for mgx_attn_fwd_log2
};

// Sum over the 64 rotation pairs of a head (lane = pair): the butterfly in the bit order 0, 1, 4, 2, 3, 5 of the pair index --
// the order in which the GEMM epilogue that does this kernel's work on the tile (gemm.hip, EPI_QKNORM) meets the pairs: four per
// lane and feature group, two groups per lane, four lanes, two waves.  Same tree, same bits.
__device__ __forceinline__ float head_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

template <bool EMIT_T>
__global__ void __launch_bounds__(256) qk_norm_rope_fwd_kernel(QkArgs a) {
  __shared__ bf16_raw vt[64][130];
  __shared__ bf16_raw qts[EMIT_T ? 64 : 1][130];
  __shared__ bf16_raw kts[EMIT_T ? 64 : 1][130];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int hh = blockIdx.y, b = blockIdx.z;
  const int t0 = blockIdx.x * 64;
  const int HD = 128;
  const long dmodel = (long)a.H * HD;
  const bool has_v = a.Vt != nullptr;             // null: V^T came straight from the projection (mgx_linear_bf16_t)
  const float wq0 = a.wq[2 * lane] * a.q_scale, wq1 = a.wq[2 * lane + 1] * a.q_scale;   // (x 1 is exact)
  const float wk0 = a.wk[2 * lane], wk1 = a.wk[2 * lane + 1];
  // four tokens per pass, every load of the pass issued before the first use: a wave instruction of this kernel moves 256 bytes
  // (one head row), so the bytes a wave keeps in flight are what its dependent chain allows -- one token at a time (three
  // loads, two reductions, two stores, sixteen times over) was latency-bound at 4.1-4.7 TB/s (profiles/r04_trainstep_kernel_stats.csv)
  constexpr int U = 4;
  for (int i0 = 0; i0 < 16; i0 += U) {
    uint32_t uq[U], uk[U], uv[U];
    float2 cc[U], ss[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + w * 16 + i0 + u;
      ok[u] = t < a.rows_per_batch;
      const int tc = ok[u] ? t : a.rows_per_batch - 1;             // clamped: loaded unconditionally, never stored
      const long row = (long)b * a.rows_per_batch + tc;
      const bf16_raw* base = a.qkv + row * a.ld + hh * HD + 2 * lane;
      uq[u] = *reinterpret_cast<const uint32_t*>(base);
      uk[u] = *reinterpret_cast<const uint32_t*>(base + dmodel);
      uv[u] = has_v ? *reinterpret_cast<const uint32_t*>(base + 2 * dmodel) : 0u;
      const long so = (long)(a.s0 + tc) * HD + 2 * lane;
      cc[u] = *reinterpret_cast<const float2*>(a.cos + so);
      ss[u] = *reinterpret_cast<const float2*>(a.sin + so);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ok[u]) continue;                                        // wave-uniform
      const int tl = w * 16 + i0 + u;
      const int s = a.s0 + t0 + tl;
      if (has_v) *reinterpret_cast<uint32_t*>(&vt[tl][2 * lane]) = uv[u];
      const float c0 = cc[u].x, c1 = cc[u].y, s0 = ss[u].x, s1 = ss[u].y;
      const long o = (((long)b * a.H + hh) * a.S + s) * HD + 2 * lane;
      {
        const float x0 = bf2f(uq[u] & 0xffff), x1 = bf2f(uq[u] >> 16);
        const float r = rsqrtf(head_sum(x0 * x0 + x1 * x1) / 128.f + 1e-6f);
        const float y0 = x0 * r * wq0, y1 = x1 * r * wq1;
        const uint32_t pk = (uint32_t)f2bf(y0 * c0 - y1 * s0) | ((uint32_t)f2bf(y1 * c1 + y0 * s1) << 16);
        *reinterpret_cast<uint32_t*>(a.Q + o) = pk;
        if (EMIT_T) *reinterpret_cast<uint32_t*>(&qts[tl][2 * lane]) = pk;
      }
      {
        const float x0 = bf2f(uk[u] & 0xffff), x1 = bf2f(uk[u] >> 16);
        const float r = rsqrtf(head_sum(x0 * x0 + x1 * x1) / 128.f + 1e-6f);
        const float y0 = x0 * r * wk0, y1 = x1 * r * wk1;
        const uint32_t pk = (uint32_t)f2bf(y0 * c0 - y1 * s0) | ((uint32_t)f2bf(y1 * c1 + y0 * s1) << 16);
        *reinterpret_cast<uint32_t*>(a.K + o) = pk;
        if (EMIT_T) *reinterpret_cast<uint32_t*>(&kts[tl][2 * lane]) = pk;
      }
      if (EMIT_T) *reinterpret_cast<uint32_t*>(a.V + o) = uv[u];
    }
  }
  if (!EMIT_T && !has_v) return;                  // (uniform over the grid)
  __syncthreads();
  // transposed tiles: Xt[b, h, d, s0 + t0 + 0..63] <- tile[t][d]
  const int ntok = min(64, a.rows_per_batch - t0);
  const int nmat = EMIT_T ? 3 : 1;
  for (int mat = 0; mat < nmat; ++mat) {
    bf16_raw (*tile)[130] = mat == 0 ? vt : (mat == 1 ? qts : kts);
    bf16_raw* outp = mat == 0 ? a.Vt : (mat == 1 ? a.Qt : a.Kt);
    for (int id = threadIdx.x; id < 128 * 8; id += 256) {
      const int d = id >> 3, c = id & 7;   // 8 tokens per 16-byte chunk
      bf16_raw* dst = outp + (((long)b * a.H + hh) * HD + d) * a.Sp + a.s0 + t0 + c * 8;
      if (c * 8 + 8 <= ntok && (((a.s0 + t0) & 7) == 0)) {
        uint4 u;
        u.x = (uint32_t)tile[c * 8 + 0][d] | ((uint32_t)tile[c * 8 + 1][d] << 16);
        u.y = (uint32_t)tile[c * 8 + 2][d] | ((uint32_t)tile[c * 8 + 3][d] << 16);
        u.z = (uint32_t)tile[c * 8 + 4][d] | ((uint32_t)tile[c * 8 + 5][d] << 16);
        u.w = (uint32_t)tile[c * 8 + 6][d] | ((uint32_t)tile[c * 8 + 7][d] << 16);
        *reinterpret_cast<uint4*>(dst) = u;
      } else {
        for (int j = 0; j < 8; ++j)
          if (c * 8 + j < ntok) dst[j] = tile[c * 8 + j][d];
      }
    }
  }
}

// backward: dQ, dK, dV [B,H,S,128] -> dqkv [rows, 3*H*128]; per-block partial weight grads
struct QkBwdArgs {
  const bf16_raw* qkv;
  long ld;
  const float* wq;
  const float* wk;
  const float* cos;
  const float* sin;
  const bf16_raw* dQ;
  const bf16_raw* dK;
  const bf16_raw* dV;
  bf16_raw* dqkv;
  long ldo;        // elements between consecutive rows of dqkv (>= 3*H*128: e.g. the [M, 7d] staging matrix of the single blocks)
  float* part;     // [nblocks][2][128]
  int H, S, Sp, rows_per_batch, s0;
  float q_scale;   // dQ is the gradient of bf16(q * q_scale): it enters multiplied by q_scale
};

__global__ void __launch_bounds__(256) qk_norm_rope_bwd_kernel(QkBwdArgs a) {
  __shared__ float red[4][2][128];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int hh = blockIdx.y, b = blockIdx.z;
  const int t0 = blockIdx.x * 64;
  const int HD = 128;
  const long dmodel = (long)a.H * HD;
  const float wq0 = a.wq[2 * lane], wq1 = a.wq[2 * lane + 1];
  const float wk0 = a.wk[2 * lane], wk1 = a.wk[2 * lane + 1];
  float gq0 = 0.f, gq1 = 0.f, gk0 = 0.f, gk1 = 0.f;
  // four tokens per pass, all loads of the pass issued before the first use (as in the forward kernel): same arithmetic, same
  // order of the weight-gradient sums
  constexpr int U = 4;
  for (int i0 = 0; i0 < 16; i0 += U) {
    uint32_t ux[U][2], ug[U][2], uv[U];
    float2 cc[U], ss[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + w * 16 + i0 + u;
      ok[u] = t < a.rows_per_batch;
      const int tc = ok[u] ? t : a.rows_per_batch - 1;
      const long row = (long)b * a.rows_per_batch + tc;
      const int s = a.s0 + tc;
      const bf16_raw* base = a.qkv + row * a.ld + hh * HD + 2 * lane;
      const long o = (((long)b * a.H + hh) * a.S + s) * HD + 2 * lane;
      ux[u][0] = *reinterpret_cast<const uint32_t*>(base);
      ux[u][1] = *reinterpret_cast<const uint32_t*>(base + dmodel);
      ug[u][0] = *reinterpret_cast<const uint32_t*>(a.dQ + o);
      ug[u][1] = *reinterpret_cast<const uint32_t*>(a.dK + o);
      uv[u] = *reinterpret_cast<const uint32_t*>(a.dV + o);
      cc[u] = *reinterpret_cast<const float2*>(a.cos + (long)s * HD + 2 * lane);
      ss[u] = *reinterpret_cast<const float2*>(a.sin + (long)s * HD + 2 * lane);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ok[u]) continue;                                        // wave-uniform
      const int t = t0 + w * 16 + i0 + u;
      const long row = (long)b * a.rows_per_batch + t;
      bf16_raw* obase = a.dqkv + row * a.ldo + hh * HD + 2 * lane;
      const float c0 = cc[u].x, c1 = cc[u].y, s0 = ss[u].x, s1 = ss[u].y;
#pragma unroll
      for (int which = 0; which < 2; ++which) {
        const float w0 = which ? wk0 : wq0, w1 = which ? wk1 : wq1;
        const float x0 = bf2f(ux[u][which] & 0xffff), x1 = bf2f(ux[u][which] >> 16);
        const float gs = which ? 1.f : a.q_scale;
        const float go0 = bf2f(ug[u][which] & 0xffff) * gs, go1 = bf2f(ug[u][which] >> 16) * gs;
        // rope^T: out0 = y0 c0 - y1 s0 ; out1 = y1 c1 + y0 s1
        const float gy0 = go0 * c0 + go1 * s1;
        const float gy1 = -go0 * s0 + go1 * c1;
        const float r = rsqrtf(head_sum(x0 * x0 + x1 * x1) / 128.f + 1e-6f);       // the forward's r, bit for bit
        // y = x * r * w
        if (which) { gk0 += gy0 * x0 * r; gk1 += gy1 * x1 * r; } else { gq0 += gy0 * x0 * r; gq1 += gy1 * x1 * r; }
        const float gz0 = gy0 * w0, gz1 = gy1 * w1;           // grad wrt (x*r)
        const float dot = wave_sum(gz0 * x0 + gz1 * x1) / 128.f;
        const float dx0 = r * gz0 - x0 * r * r * r * dot;
        const float dx1 = r * gz1 - x1 * r * r * r * dot;
        *reinterpret_cast<uint32_t*>(obase + which * dmodel) = (uint32_t)f2bf(dx0) | ((uint32_t)f2bf(dx1) << 16);
      }
      *reinterpret_cast<uint32_t*>(obase + 2 * dmodel) = uv[u];
    }
  }
  red[w][0][2 * lane] = gq0; red[w][0][2 * lane + 1] = gq1;
  red[w][1][2 * lane] = gk0; red[w][1][2 * lane + 1] = gk1;
  __syncthreads();
  const long blk = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  {
    const int which = threadIdx.x >> 7, c = threadIdx.x & 127;
    a.part[(blk * 2 + which) * 128 + c] = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
  }
}

// sum the per-block partials in two stages (a single stage of 16 workgroups walking 12,096 partial rows took 174 us):
// stage 1: workgroup g of G sums the rows g, g + G, ... of all 256 columns into row g of `mid`; stage 2: one block per
// (which, 16-column slab), 256 threads = 16 columns x 16 row-lanes, sums the G rows of `mid` and accumulates into gq / gk.
constexpr int QK_FIN_G = 128;
__global__ void __launch_bounds__(256) qk_bwd_partial_kernel(const float* __restrict__ part, float* __restrict__ mid,
                                                             long nblocks) {
  float s = 0.f;
  for (long k = blockIdx.x; k < nblocks; k += QK_FIN_G) s += part[k * 256 + threadIdx.x];
  mid[(long)blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) qk_bwd_finish_kernel(const float* __restrict__ part, float* __restrict__ gq,
                                                            float* __restrict__ gk, long nblocks) {
  __shared__ float red[16][17];
  const int which = blockIdx.x >> 3, c0 = (blockIdx.x & 7) * 16;
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  float s = 0.f;
  for (long k = ry; k < nblocks; k += 16) s += part[(k * 2 + which) * 128 + c0 + cx];
  red[ry][cx] = s;
  __syncthreads();
  if (ry == 0) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i][cx];
    float* dst = which ? gk : gq;
    dst[c0 + cx] += t;
  }
}

}  // namespace

extern "C" int mgx_ln_modulate_fwd(const uint16_t* x, long ldx, long x_rpb, long x_bstride, const uint16_t* shift,
                                   const uint16_t* scale, long mod_ld, uint16_t* y, long ldy, float* stats, long M, int D,
                                   void* stream) {
  MGX_REQUIRE(x && shift && scale && y && M > 0, "bad argument");
  MGX_REQUIRE(D % 512 == 0 && D <= 4096, "feature size must be a multiple of 512 up to 4096");
  MGX_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && mod_ld % 8 == 0 && x_bstride % 8 == 0, "16-byte row alignment");
  hipStream_t st = (hipStream_t)stream;
  const RowMap xm{ldx, x_rpb, x_bstride};
  const int grid = cdiv(M, 4);
  switch (D / 512) {
#define CASE(nv) case nv: ln_mod_fwd_kernel<nv><<<grid, 256, 0, st>>>(x, xm, shift, scale, mod_ld, y, ldy, stats, M); break;
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
  }
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" long mgx_ln_modulate_bwd_workspace(long M, long rpb, int D) {
  const long batches = (M + rpb - 1) / rpb;
  const long bpb = ((rpb < M ? rpb : M) + LN_BWD_ROWS - 1) / LN_BWD_ROWS;
  return batches * bpb * 2 * D;
}

extern "C" int mgx_ln_modulate_bwd(const uint16_t* dy, long lddy, const uint16_t* x, long ldx, long x_rpb, long x_bstride,
                                   const uint16_t* scale, long mod_ld, uint16_t* dx, long lddx, long dx_rpb,
                                   long dx_bstride, int accumulate, uint16_t* dshift, uint16_t* dscale, float* ws, long M,
                                   int D, void* stream) {
  MGX_REQUIRE(dy && x && scale && dx && dshift && dscale && ws && M > 0, "bad argument");
  MGX_REQUIRE(D % 512 == 0 && D <= 4096, "feature size must be a multiple of 512 up to 4096");
  MGX_REQUIRE(x_rpb == dx_rpb, "x and dx must be batched alike");
  hipStream_t st = (hipStream_t)stream;
  const long rpb = x_rpb < M ? x_rpb : M;
  const int batches = cdiv(M, rpb);
  const int bpb = cdiv(rpb, LN_BWD_ROWS);
  const RowMap xm{ldx, rpb, x_bstride}, dxm{lddx, rpb, dx_bstride};
  const size_t lds = (size_t)2 * D * sizeof(float);
  switch (D / 512) {
#define CASE(nv)                                                                                                  \
  case nv:                                                                                                        \
    ln_mod_bwd_kernel<nv><<<batches * bpb, 256, lds, st>>>(dy, lddy, x, xm, scale, mod_ld, dx, dxm, accumulate, ws, M, bpb); \
    break;
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
  }
  ln_mod_bwd_finish_kernel<<<dim3(cdiv(D, 64), batches), 256, 0, st>>>(ws, dshift, dscale, mod_ld, D, bpb);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int mgx_qk_norm_rope_fwd_qs(const uint16_t* qkv, long ld, const float* wq, const float* wk, const float* cos,
                                       const float* sin, uint16_t* Q, uint16_t* K, uint16_t* Vt, uint16_t* V, uint16_t* Qt,
                                       uint16_t* Kt, int B, int H, int S, int Sp, int rows_per_batch, int s0, float q_scale,
                                       void* stream) {
  MGX_REQUIRE(qkv && wq && wk && cos && sin && Q && K, "null argument");
  MGX_REQUIRE((V != nullptr) == (Qt != nullptr) && (V != nullptr) == (Kt != nullptr), "V, Qt, Kt come together");
  MGX_REQUIRE(Vt || !V, "Vt may only be left out (V^T written by mgx_linear_bf16_t) without the backward's extra layouts");
  MGX_REQUIRE(B > 0 && H > 0 && rows_per_batch > 0 && s0 >= 0 && s0 + rows_per_batch <= S && Sp >= S, "bad sizes");
  MGX_REQUIRE(ld == 3L * H * 128, "qkv rows must be [q | k | v] of H*128 each");
  MGX_REQUIRE(q_scale > 0.f, "q_scale must be positive");
  QkArgs a{qkv, ld, wq, wk, cos, sin, Q, K, Vt, V, Qt, Kt, H, S, Sp, rows_per_batch, s0, q_scale};
  dim3 grid(cdiv(rows_per_batch, 64), H, B);
  if (V) qk_norm_rope_fwd_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(a);
  else qk_norm_rope_fwd_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(a);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int mgx_qk_norm_rope_fwd(const uint16_t* qkv, long ld, const float* wq, const float* wk, const float* cos,
                                    const float* sin, uint16_t* Q, uint16_t* K, uint16_t* Vt, uint16_t* V, uint16_t* Qt,
                                    uint16_t* Kt, int B, int H, int S, int Sp, int rows_per_batch, int s0, void* stream) {
  return mgx_qk_norm_rope_fwd_qs(qkv, ld, wq, wk, cos, sin, Q, K, Vt, V, Qt, Kt, B, H, S, Sp, rows_per_batch, s0, 1.0f, stream);
}

extern "C" long mgx_qk_norm_rope_bwd_workspace(int B, int H, int rows_per_batch) {
  return ((long)cdiv(rows_per_batch, 64) * H * B + QK_FIN_G) * 2 * 128;    // per-block partials + the first stage's sums
}

extern "C" int mgx_qk_norm_rope_bwd_qs(const uint16_t* qkv, long ld, const float* wq, const float* wk, const float* cos,
                                       const float* sin, const uint16_t* dQ, const uint16_t* dK, const uint16_t* dV,
                                       uint16_t* dqkv, long ld_dqkv, float* gwq, float* gwk, float* ws, int B, int H, int S,
                                       int Sp, int rows_per_batch, int s0, float q_scale, void* stream) {
  MGX_REQUIRE(qkv && wq && wk && cos && sin && dQ && dK && dV && dqkv && gwq && gwk && ws, "null argument");
  MGX_REQUIRE(ld == 3L * H * 128, "qkv rows must be [q | k | v] of H*128 each");
  MGX_REQUIRE(ld_dqkv >= ld && ld_dqkv % 2 == 0, "dqkv rows must hold [dq | dk | dv] and keep 4-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  QkBwdArgs a{qkv, ld, wq, wk, cos, sin, dQ, dK, dV, dqkv, ld_dqkv, ws, H, S, Sp, rows_per_batch, s0, q_scale};
  dim3 grid(cdiv(rows_per_batch, 64), H, B);
  qk_norm_rope_bwd_kernel<<<grid, 256, 0, st>>>(a);
  const long nblocks = (long)grid.x * grid.y * grid.z;
  float* mid = ws + nblocks * 256;
  qk_bwd_partial_kernel<<<QK_FIN_G, 256, 0, st>>>(ws, mid, nblocks);
  qk_bwd_finish_kernel<<<16, 256, 0, st>>>(mid, gwq, gwk, QK_FIN_G);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int mgx_qk_norm_rope_bwd(const uint16_t* qkv, long ld, const float* wq, const float* wk, const float* cos,
                                    const float* sin, const uint16_t* dQ, const uint16_t* dK, const uint16_t* dV,
                                    uint16_t* dqkv, long ld_dqkv, float* gwq, float* gwk, float* ws, int B, int H, int S,
                                    int Sp, int rows_per_batch, int s0, void* stream) {
  return mgx_qk_norm_rope_bwd_qs(qkv, ld, wq, wk, cos, sin, dQ, dK, dV, dqkv, ld_dqkv, gwq, gwk, ws, B, H, S, Sp, rows_per_batch,
                                 s0, 1.0f, stream);
}
