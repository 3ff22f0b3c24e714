// bf16 MFMA GEMM for the MMDiT linears (gfx950):  C[M,N] = epi( A[M,K] @ W[N,K]^T + bias )
//
// Both operands are K-contiguous ("NT"): activations [tokens, features] and torch Linear weights [out, in].
// Two kernels: `gemm_pp_kernel` (256x256x64 tiles, persistent, LDS-DMA staging, ping-pong K-loop: every problem with >= 128
// such tiles, i.e. all the FLOPs that matter; described at its definition) and `gemm_kernel` for the small problems:
// tile 128x128x64, 256 threads = 4 waves (2x2), each wave a 64x64 sub-tile as 4x4 v_mfma_f32_16x16x32_bf16
// accumulators.  Operands are staged global -> registers -> LDS (issue-early / write-late, one barrier per
// K-tile, two LDS buffers); the LDS image is [row][64 k] with an XOR swizzle on the 16-byte chunk index
// (chunk ^ ((row>>1)&7)) which makes every ds_read_b128 fragment read and every ds_write_b128 conflict-free.
// The MFMA is issued with the weight fragment as the A operand so that each lane ends up with 4 consecutive
// output features of one token: 8-byte bf16 stores, and bias/gate vectors are per-register constants.
// blockIdx is remapped so that the 8 XCDs each own a contiguous band of tiles (private L2 reuse of W/A panels).
//
// "Row-batched" addressing (rpb, bstride) lets the text and image streams live inside one joint
// [B, S, d] residual buffer: row m -> base + (m / rpb) * bstride + (m % rpb) * ld.
#include "../../include/mixgrpo_hip.h"
#include "common.h"

#include <cmath>
#include <cstdlib>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NT = 256;

struct RowMap {
  long ld;       // elements between consecutive rows inside a batch
  long rpb;      // rows per batch (>= M for a plain matrix; the host clamps it to 2^30 so 32-bit division applies)
  long bstride;  // elements between batches
};

__device__ __forceinline__ long row_off(const RowMap& r, long m) {
  const uint32_t rpb = (uint32_t)r.rpb;
  const uint32_t b = (uint32_t)m / rpb;         // M < 2^31 and rpb <= 2^30: a 32-bit division (a 64-bit one is ~120 instructions)
  return (long)b * r.bstride + (long)((uint32_t)m - b * rpb) * r.ld;
}

enum Epi { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_GATE_RES = 2, EPI_F32_ACC = 3, EPI_BIAS_MULAUX = 4,
           EPI_QKNORM = 5 /* internal (mgx_linear_qk_norm_rope): bias, per-head RMSNorm, RoPE, head split */,
           EPI_QKNORM_P = 6 /* the same reading a (cos, sin)-per-pair table: a kernel of its own -- a run-time choice inside the
                               epilogue sends its register arrays to scratch */ };

struct GemmArgs {
  const bf16_raw* A;
  const bf16_raw* W;
  const bf16_raw* bias;   // [N] or null
  void* C;                // bf16 (or fp32 for EPI_F32_ACC)
  const bf16_raw* gate;   // [batches, gate_ld] (EPI_BIAS_GATE_RES): gate[b*gate_ld + n]
  bf16_raw* aux;          // optional second output / input: plain matrix, row m at aux + m*ldaux
  long ldaux;
  long gate_ld;
  int M, N, K;
  RowMap a, c;
  long ldw;
  float beta;             // EPI_F32_ACC: C = beta*C + acc
  int rowwise_ok;         // all bf16 side operands are 16-byte addressable: the LDS-staged epilogue may be used
  int span32;             // both operands span < 4 GiB: the persistent kernel's 32-bit DMA source offsets are valid
  int band;               // tile order of the persistent kernel (set in launch(): rule at gemm_pp_kernel)
  // implicit 3x3 convolution over a zero-bordered NHWC image (mgx_conv3x3_nhwc): K = 9 taps x C channels, K-tile kt is
  // channels 64 (kt % cpt) .. of tap kt / cpt, cpt = C / 64 = 1 << conv_shift; its A rows start conv_dy elements per tap
  // row and conv_dx per tap column further on.  conv_shift < 0: a plain GEMM (K contiguous).
  int conv_shift;
  long conv_dy, conv_dx;
  // stream-K tail of the persistent kernel (described at gemm_pp_kernel): sk_ws = fp32 workspace for partial tiles (null: off),
  // sk_minparts = split an XCD's last, partial round only if each of its tiles can then be cut into at least this many parts
  float* sk_ws;
  int sk_minparts;
  // second problem of a PAIR launch (mgx_gemm_bf16_pair; the text- and image-stream Linear of a double block): rows
  // m_split.. of the tile grid (m_split % 256 == 0; 0: no pair) are rows 0.. of a problem with its own operands.  Same N, K,
  // epilogue and leading dimensions.  A2 / W2 are reached through 32-bit offsets from A / W (both streams' operands live in the
  // same buffers), so the K-loop is the single-problem loop; only the per-tile offsets and the epilogue's operands differ.
  int m_split;
  long a1_off, w1_off;    // elements from A / W (= the lower of the two problems' pointers) to problem 1's first element
  long a2_off, w2_off;    // ... and to problem 2's
  RowMap a2, c2;
  void* C2;
  const bf16_raw* bias2;
  const bf16_raw* gate2;
  bf16_raw* aux2;
  // transposed-output Linear (mgx_linear_bf16_t; EPI_BIAS, persistent kernel, 16-byte epilogue only): the roles are swapped -- A =
  // the weight rows (M = output features), W = the activations (N = tokens) -- so C[feature][token] comes out token-contiguous:
  // V^T of the attention without a transposing pass.  bias_rows: bias[M] is indexed by the ROW; col_rpb > 0: column n lies in
  // column batch n / col_rpb at C + (n / col_rpb) * col_bstride + row offset + n % col_rpb (col_rpb % 64 == 0).
  int bias_rows;
  long col_rpb, col_bstride;
  // EPI_QKNORM (mgx_linear_qk_norm_rope; persistent kernel only): the output columns are [q | k] of qn_dmodel = H * 128 each
  // (qn_dmodel % 256 == 0: a tile lies in one section, a wave's 64 columns in one half of one head); c.rpb = tokens per sample
  const float* qn_wq;     // [128] fp32 RMSNorm weights
  const float* qn_wk;
  const float* qn_cos;    // [S, 128] fp32
  const float* qn_sin;
  const float* qn_cs2;    // EPI_QKNORM_P: [S, 64, 2] fp32 (cos, sin) per rotation pair, for tables whose two entries of a pair are
                          // equal (FLUX's): half the epilogue's table bytes
  bf16_raw* qn_Q;         // [B, H, S, 128]
  bf16_raw* qn_K;
  int qn_H, qn_S, qn_s0, qn_dmodel;
  float qn_qscale;
};

// operands of the problem that tile row m0 belongs to (PAIR launches), as a GemmArgs the ordinary epilogue can take
template <bool PAIR>
__device__ __forceinline__ GemmArgs seg_args(const GemmArgs& g, long& m0) {
  if (!PAIR || g.m_split == 0 || m0 < g.m_split) {
    GemmArgs r = g;
    if (PAIR && g.m_split) r.M = g.m_split;
    return r;
  }
  GemmArgs r = g;
  r.C = g.C2; r.c = g.c2; r.bias = g.bias2; r.gate = g.gate2; r.aux = g.aux2;
  r.M = g.M - g.m_split;
  m0 -= g.m_split;
  return r;
}

// element offset of K-tile kt inside an A row
template <bool CONV>
__device__ __forceinline__ long a_koff(const GemmArgs& g, int kt) {
  if (!CONV) return (long)kt * BK;
  const int tap = kt >> g.conv_shift, c = kt - (tap << g.conv_shift);
  const int ty = (tap * 11) >> 5;                        // tap / 3 for tap < 9
  return (long)ty * g.conv_dy + (long)(tap - 3 * ty) * g.conv_dx + (long)c * BK;
}

// gelu_tanh(x) = 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3).  With tanh(u) = 2 s - 1, s = 1 / (1 + e^{-2u}):
// gelu = x s and gelu' = s + 2 x s (1 - s) u'.  One v_exp_f32 + one v_rcp_f32 per element instead of the ~40-instruction
// ocml tanhf (the epilogue of a 256x256 tile evaluates it 65,536 times; both agree to < 1 ulp of the bf16 result).
__device__ __forceinline__ float gelu_sigmoid_f(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float c = -2.0f * k0 * 1.4426950408889634f;            // e^{-2u} = 2^{c x (1 + k1 x^2)}
  const float e = __builtin_amdgcn_exp2f(c * x * __builtin_fmaf(k1 * x, x, 1.0f));
  return __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float gelu_tanh_f(float x) { return x * gelu_sigmoid_f(x); }
__device__ __forceinline__ float gelu_tanh_grad_f(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float s = gelu_sigmoid_f(x);
  const float du = k0 * __builtin_fmaf(3.0f * k1 * x, x, 1.0f);
  return __builtin_fmaf(2.0f * x * s * (1.0f - s), du, s);
}

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// single-instruction fp32 ops the SLP vectoriser cannot pack into v_pk_*_f32 (MI355X_MICROARCH.md: packed fp32 beside MFMAs is an
// anti-lever; the two waves of a SIMD run their epilogues beside each other's MFMA sections)
__device__ __forceinline__ float vfma1(float sa /* wave-uniform */, float b, float c) { float r; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "s"(sa), "v"(b), "v"(c)); return r; }

// Epilogue of the 128x128 kernel: lane holds, for m-tile j and n-tile i, 4 consecutive features of one token.
template <int EPI, int MT, int NTL>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x4 (&acc)[NTL][MT], int lane, long m0, long n0, int wm,
                                              int wn) {
  const int fr = lane & 15, fq = lane >> 4;
  // token m = m0 + wm*64 + j*16 + fr, features n .. n+3
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const long m = m0 + wm * (MT * 16) + j * 16 + fr;
    if (m >= g.M) continue;
    const long crow = row_off(g.c, m);
    const long bidx = (uint32_t)m / (uint32_t)g.c.rpb;
#pragma unroll
    for (int i = 0; i < NTL; ++i) {
      const long n = n0 + wn * 64 + i * 16 + fq * 4;
      if (n >= g.N) continue;   // N is a multiple of 4 (checked on the host)
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (EPI == EPI_F32_ACC) {
        float4* cp = reinterpret_cast<float4*>(reinterpret_cast<float*>(g.C) + crow + n);
        float4 o = make_float4(v[0], v[1], v[2], v[3]);
        if (g.beta != 0.f) {
          const float4 old = *cp;
          o.x += g.beta * old.x; o.y += g.beta * old.y; o.z += g.beta * old.z; o.w += g.beta * old.w;
        }
        *cp = o;
        continue;
      }
      if (g.bias) {
        const uint2 bb = *reinterpret_cast<const uint2*>(g.bias + n);
        v[0] += bf2f(bb.x & 0xffff); v[1] += bf2f(bb.x >> 16); v[2] += bf2f(bb.y & 0xffff); v[3] += bf2f(bb.y >> 16);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = rbf(v[r]);   // the Linear's bf16 output
      bf16_raw* cp = reinterpret_cast<bf16_raw*>(g.C) + crow + n;
      if (EPI == EPI_BIAS_GELU) {
        if (g.aux) {  // keep the pre-activation for the backward pass
          uint2 pre;
          pre.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
          pre.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
          *reinterpret_cast<uint2*>(g.aux + m * g.ldaux + n) = pre;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_f(v[r]);
      } else if (EPI == EPI_BIAS_GATE_RES) {
        if (g.aux) {  // pre-gate branch output, needed for d(gate)
          uint2 pre;
          pre.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
          pre.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
          *reinterpret_cast<uint2*>(g.aux + m * g.ldaux + n) = pre;
        }
        const uint2 gg = *reinterpret_cast<const uint2*>(g.gate + bidx * g.gate_ld + n);
        const uint2 rr = *reinterpret_cast<const uint2*>(cp);
        v[0] = bf2f(rr.x & 0xffff) + rbf(bf2f(gg.x & 0xffff) * v[0]);
        v[1] = bf2f(rr.x >> 16) + rbf(bf2f(gg.x >> 16) * v[1]);
        v[2] = bf2f(rr.y & 0xffff) + rbf(bf2f(gg.y & 0xffff) * v[2]);
        v[3] = bf2f(rr.y >> 16) + rbf(bf2f(gg.y >> 16) * v[3]);
      } else if (EPI == EPI_BIAS_MULAUX) {
        // dgrad through GELU: C = (A@W^T) * gelu'(aux)   (aux = saved pre-activation)
        const uint2 pp = *reinterpret_cast<const uint2*>(g.aux + m * g.ldaux + n);
        v[0] *= gelu_tanh_grad_f(bf2f(pp.x & 0xffff)); v[1] *= gelu_tanh_grad_f(bf2f(pp.x >> 16));
        v[2] *= gelu_tanh_grad_f(bf2f(pp.y & 0xffff)); v[3] *= gelu_tanh_grad_f(bf2f(pp.y >> 16));
      }
      uint2 o;
      o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
      o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
      *reinterpret_cast<uint2*>(cp) = o;
    }
  }
}

template <int EPI, bool CONV>
__global__ void __launch_bounds__(256, 2) gemm_kernel(GemmArgs g) {
  constexpr int TM = 128, TN = 128;                               // block tile
  constexpr int NTHR = 256;
  constexpr int RS = NTHR / 8;                                    // rows covered by one pass of the loader
  constexpr int TB = TM * 128;                                    // bytes of one operand tile (64 k x 2 B per row)
  constexpr int STAGE = 2 * TB;
  constexpr int MT = 4, NTL = 4;                                  // 16x16 tiles per wave along M / N
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;

  // XCD-aware tile order: consecutive logical tiles (which share A/W panels) go to the same XCD
  const int tiles_m = (g.M + TM - 1) / TM, tiles_n = (g.N + TN - 1) / TN;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // walk N fastest inside a band of 8 M-tiles so a wave of tiles reuses the same W panels and A panels
  const int band = 8;
  const int per_band = band * tiles_n;
  const int b0 = bid / per_band;
  const int rows_in_band = min(band, tiles_m - b0 * band);
  const int in_band = bid - b0 * per_band;
  const int tm = b0 * band + in_band % rows_in_band;
  const int tn = in_band / rows_in_band;
  const long m0 = (long)tm * TM, n0 = (long)tn * TN;

  // ---- loader: thread owns 16-byte chunks (row = lrow + RS*i, kc = lkc), i = 0..3; RS*i leaves bits 1..3 of the row
  // alone, so the swizzle term is the same for all four and one LDS offset (+ RS*128*i) serves them
  const int lrow = tid >> 3, lkc = tid & 7;
  const int src_kc = lkc;
  auto a_ptr = [&](int i) {
    long m = m0 + lrow + RS * i;
    if (m >= g.M) m = g.M - 1;
    return g.A + row_off(g.a, m) + src_kc * 8;
  };
  auto w_ptr = [&](int i) {
    long n = n0 + lrow + RS * i;
    if (n >= g.N) n = g.N - 1;
    return g.W + n * g.ldw + src_kc * 8;
  };
  const bf16_raw* ap0 = a_ptr(0); const bf16_raw* ap1 = a_ptr(1); const bf16_raw* ap2 = a_ptr(2); const bf16_raw* ap3 = a_ptr(3);
  const bf16_raw* wp0 = w_ptr(0); const bf16_raw* wp1 = w_ptr(1); const bf16_raw* wp2 = w_ptr(2); const bf16_raw* wp3 = w_ptr(3);
  const int lds0 = lrow * 128 + swz(lrow, lkc) * 16;
  uint4 ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3;
#define LOAD_TILE(kt)                                                            \
  do {                                                                           \
    const long ko = (long)(kt) * BK;                                             \
    const long ka = a_koff<CONV>(g, (kt));                                       \
    ra0 = *reinterpret_cast<const uint4*>(ap0 + ka);                             \
    ra1 = *reinterpret_cast<const uint4*>(ap1 + ka);                             \
    ra2 = *reinterpret_cast<const uint4*>(ap2 + ka);                             \
    ra3 = *reinterpret_cast<const uint4*>(ap3 + ka);                             \
    rw0 = *reinterpret_cast<const uint4*>(wp0 + ko);                             \
    rw1 = *reinterpret_cast<const uint4*>(wp1 + ko);                             \
    rw2 = *reinterpret_cast<const uint4*>(wp2 + ko);                             \
    rw3 = *reinterpret_cast<const uint4*>(wp3 + ko);                             \
  } while (0)
#define STORE_TILE(buf)                                                          \
  do {                                                                           \
    char* base_ = smem + (buf) * STAGE + lds0;                                   \
    *reinterpret_cast<uint4*>(base_) = ra0;                                      \
    *reinterpret_cast<uint4*>(base_ + RS * 128) = ra1;                           \
    *reinterpret_cast<uint4*>(base_ + 2 * RS * 128) = ra2;                       \
    *reinterpret_cast<uint4*>(base_ + 3 * RS * 128) = ra3;                       \
    *reinterpret_cast<uint4*>(base_ + TB) = rw0;                                 \
    *reinterpret_cast<uint4*>(base_ + TB + RS * 128) = rw1;                      \
    *reinterpret_cast<uint4*>(base_ + TB + 2 * RS * 128) = rw2;                  \
    *reinterpret_cast<uint4*>(base_ + TB + 3 * RS * 128) = rw3;                  \
  } while (0)

  f32x4 acc[NTL][MT];  // [n-tile][m-tile]
#pragma unroll
  for (int i = 0; i < NTL; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  const int nkt = g.K / BK;
  LOAD_TILE(0);
  STORE_TILE(0);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) LOAD_TILE(kt + 1);
    const char* sa = smem + cur * STAGE;
    const char* sw = sa + TB;
    // all fragments of the K-tile are requested up front (both 32-deep k-steps): the second k-step's ds_reads
    // return underneath the first k-step's MFMAs instead of in an MFMA-idle phase
    s16x8 fa[2][MT], fw[2][NTL];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        const int rw_ = wn * 64 + t * 16 + fr;
        fw[ks][t] = *reinterpret_cast<const s16x8*>(sw + rw_ * 128 + swz(rw_, ks * 4 + fq) * 16);
      }
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int ra_ = wm * (MT * 16) + t * 16 + fr;
        fa[ks][t] = *reinterpret_cast<const s16x8*>(sa + ra_ * 128 + swz(ra_, ks * 4 + fq) * 16);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int i = 0; i < NTL; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ks][i], fa[ks][j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nkt) STORE_TILE(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
#undef LOAD_TILE
#undef STORE_TILE

  gemm_epilogue<EPI, MT, NTL>(g, acc, lane, m0, n0, wm, wn);
}

// ------------------------------------------------------------------------------------------ transpose
// out[n][m] = in[m][n] (bf16), in rows row-batched, out [N][ldo] with ldo >= M (zero padded up to ldo by the
// host memset).  Optionally emits per-block column partial sums (for bias gradients): part[blockIdx.y][n].
__global__ void __launch_bounds__(256) transpose_kernel(const bf16_raw* __restrict__ in, bf16_raw* __restrict__ out,
                                                        float* __restrict__ part, int M, int N, RowMap im, long ldo) {
  __shared__ bf16_raw tile[64][66];
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const long m = m0 + r;
    const int n = n0 + tx;
    bf16_raw v = 0;
    if (m < M && n < N) v = in[row_off(im, m) + n];
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int n = n0 + r;
    const long m = m0 + tx;
    if (n < N && m < ldo) out[(long)n * ldo + m] = (m < M) ? tile[tx][r] : (bf16_raw)0;
  }
  if (part && m0 < M) {
    if (ty == 0) {
      float s = 0.f;
      for (int r = 0; r < 64; ++r) s += bf2f(tile[r][tx]);
      if (n0 + tx < N) part[(long)blockIdx.y * N + n0 + tx] = s;
    }
  }
}

// Fast path (N % 8 == 0, 16-byte aligned rows): each thread moves an 8x8 block through registers (8 x 16-byte loads,
// 32 v_perm, 8 x 16-byte stores): no LDS, 128-byte segments on both sides.  Block = 16x16 threads = 128x128 tile.
__device__ __forceinline__ uint32_t lo16(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040100); }
__device__ __forceinline__ uint32_t hi16(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302); }
// GELU = true: out = transpose(bf16(gelu_tanh(float(in)))) -- the activation a KEPT FF pre-activation stands for, formed on the
// way into the weight-gradient operand instead of by a pass of its own (`gelu_rows_kernel` + this kernel; bit-identical values).
template <bool GELU>
__global__ void __launch_bounds__(256) transpose8_kernel(const bf16_raw* __restrict__ in, bf16_raw* __restrict__ out,
                                                         float* __restrict__ part, int M, int N, RowMap im, long ldo) {
  __shared__ float cs[16][129];
  // tx: 8-column group, ty: 8-row group.  A wave covers 8 x 8 of them (64 columns x 64 rows), so that every load
  // instruction of the wave reads 8 rows x 128 contiguous bytes and every store instruction writes 8 rows x 128 bytes
  // (with 16 x 4 groups per wave the stores were 64-byte pieces of 16 different rows; profiles/r02_transpose_ab.log).
  const int tx = (threadIdx.x & 7) | (((threadIdx.x >> 6) & 1) << 3), ty = ((threadIdx.x >> 3) & 7) | ((threadIdx.x >> 7) << 3);
  const int n0 = blockIdx.x * 128 + tx * 8;
  const long m0 = (long)blockIdx.y * 128 + ty * 8;
  uint32_t r[8][4];
  float csum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) csum[j] = 0.f;
  const bool ncol_ok = n0 < N;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const long m = m0 + i;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (ncol_ok && m < M) v = *reinterpret_cast<const uint4*>(in + row_off(im, m) + n0);
    if (GELU && ncol_ok && m < M) {
      auto gl = [](uint32_t u) {        // exactly what the bias+GELU epilogue and gelu_rows_kernel apply to the pre-activation
        return (uint32_t)f2bf(gelu_tanh_f(bf2f(u & 0xffff))) | ((uint32_t)f2bf(gelu_tanh_f(bf2f(u >> 16))) << 16);
      };
      v.x = gl(v.x); v.y = gl(v.y); v.z = gl(v.z); v.w = gl(v.w);
    }
    r[i][0] = v.x; r[i][1] = v.y; r[i][2] = v.z; r[i][3] = v.w;
    if (part) {
      csum[0] += bf2f(v.x & 0xffff); csum[1] += bf2f(v.x >> 16); csum[2] += bf2f(v.y & 0xffff); csum[3] += bf2f(v.y >> 16);
      csum[4] += bf2f(v.z & 0xffff); csum[5] += bf2f(v.z >> 16); csum[6] += bf2f(v.w & 0xffff); csum[7] += bf2f(v.w >> 16);
    }
  }
  // out row (n0 + c) = [in[m0+0][c], in[m0+1][c], ..., in[m0+7][c]]
  if (ncol_ok && m0 < ldo) {
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2) {           // column pair (2*c2, 2*c2+1) lives in dword c2 of every row
      uint4 e, o;
      e.x = lo16(r[0][c2], r[1][c2]); e.y = lo16(r[2][c2], r[3][c2]); e.z = lo16(r[4][c2], r[5][c2]); e.w = lo16(r[6][c2], r[7][c2]);
      o.x = hi16(r[0][c2], r[1][c2]); o.y = hi16(r[2][c2], r[3][c2]); o.z = hi16(r[4][c2], r[5][c2]); o.w = hi16(r[6][c2], r[7][c2]);
      *reinterpret_cast<uint4*>(out + (long)(n0 + 2 * c2) * ldo + m0) = e;
      *reinterpret_cast<uint4*>(out + (long)(n0 + 2 * c2 + 1) * ldo + m0) = o;
    }
  }
  if (part) {
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[ty][tx * 8 + j] = csum[j];
    __syncthreads();
    if (threadIdx.x < 128) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += cs[k][threadIdx.x];
      const int n = blockIdx.x * 128 + threadIdx.x;
      if (n < N && (long)blockIdx.y * 128 < M) part[(long)blockIdx.y * N + n] = t;
    }
  }
}

// Block = 64 columns x 4 lanes over the partial rows
__global__ void __launch_bounds__(256) colsum_finish_kernel(const float* __restrict__ part, float* __restrict__ out, int nblk,
                                                            int N, float beta) {
  __shared__ float red[4][64];
  const int cx = threadIdx.x & 63, ky = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + cx;
  float s = 0.f;
  if (n < N)
    for (int b = ky; b < nblk; b += 4) s += part[(long)b * N + n];
  red[ky][cx] = s;
  __syncthreads();
  if (ky == 0 && n < N) out[n] = beta * out[n] + ((red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]));
}

// ------------------------------------------------------------------------------------------ persistent variant
// One workgroup per CU walks a strided list of 256x256 tiles.  The first K-tile of the NEXT tile is DMA'd into the
// idle LDS buffer during the last K-step of the current one (no exposed prologue, no dispatch gap between tiles) and
// the epilogue's global stores drain underneath the next tile's K-loop.
//
// Output layout trick: the W rows of each 64-row wave group are stored in LDS in the order
//     LDS row t*16 + f  <-  W row (f>>2)*16 + t*4 + (f&3)            (the DMA source address is per lane: free)
// so the accumulators acc[t = 0..3][j] of lane (fr, fq) are 16 CONSECUTIVE output features fq*16 + t*4 + r of token
// row j*16 + fr: the epilogue runs straight from registers with 16-byte accesses (4 lanes = one 128-byte line per
// row), with no LDS transpose, no bf16 round trip and no barrier before the next tile.
//
// Everything the epilogue reads from memory is requested UNCONDITIONALLY (clamped addresses) and in batches: a
// per-row `if (in range) { load; use; store }` makes hipcc emit branch + load + s_waitcnt vmcnt(0) per row, i.e.
// serial HBM round trips (vmcnt counts stores too, so each wait also drains the previous row's store): that form
// cost ~20 us of an 80 us K=3072 tile.
// (The lane's four accumulator tiles: t = 0, 1 are features fq*8 .. fq*8+7 and tiles t = 2, 3 the same + 32, so that ONE 16-byte store instruction of the wave
// writes 64 contiguous bytes per token row -- half a 128-byte line -- instead of four 16-byte pieces 32 bytes apart, which is
// what 16 CONSECUTIVE features per lane gave: profiles/r02_ab_gemm_variants.log.)
__device__ __forceinline__ int wperm(int p) { return (p >> 5) * 32 + ((p & 15) >> 2) * 8 + ((p >> 4) & 1) * 4 + (p & 3); }
__device__ __forceinline__ int lane_feat(int fq, int t) { return (t >> 1) * 32 + fq * 8 + (t & 1) * 4; }

__device__ __forceinline__ void unpack8(const uint4& u, float* v) {
  v[0] = bf2f(u.x & 0xffff); v[1] = bf2f(u.x >> 16); v[2] = bf2f(u.y & 0xffff); v[3] = bf2f(u.y >> 16);
  v[4] = bf2f(u.z & 0xffff); v[5] = bf2f(u.z >> 16); v[6] = bf2f(u.w & 0xffff); v[7] = bf2f(u.w >> 16);
}
__device__ __forceinline__ uint4 pack8(const float* v) {
  uint4 o;
  o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
  o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
  o.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16);
  o.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
  return o;
}

// EPI_QKNORM: what mgx_qk_norm_rope_fwd does to the q | k columns, on the tile while it is in registers -- the [tokens, 2d]
// projection output is never written and never re-read (the rollout's norm pass was 0.35 s of an 18.3 s step at HBM speed).
// The arithmetic is that kernel's, operation for operation and in its summation order (norm.hip head_sum: the butterfly over
// the 64 rotation pairs of a head in the bit order 0, 1, 4, 2, 3, 5 of the pair index), on the bf16-ROUNDED Linear output, with
// no contraction: both paths give the same bits (tests/test_hip_gemm.py).  A head's 128 features are two waves' columns
// (wn, wn ^ 1): the half-head sums of squares cross through LDS with ONE workgroup barrier -- every wave of the workgroup runs
// it exactly once per tile, so the early / late halves of the K-loop stay one barrier apart.  `red`: the A stage the NEXT
// K-tile DMA will fill; a wave's sums are written where its PARTNER's first DMA piece lands (bytes (wu ^ 1) * 1024 ..), so only
// the reader itself can overwrite them, after it has read.
// single-instruction fp32 ops the SLP vectoriser cannot pack (a plain -O3 build turns the pair form's adjacent multiplies into
// v_pk_mul_f32 on register pairs and spills 200 bytes per lane around them: 1.217 ms against 1.123 ms without SLP)
__device__ __forceinline__ float vmul1(float a, float b) { float r; asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vsub1(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vadd1(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

template <bool PAIRED>
__device__ __forceinline__ void qknorm_epilogue(const GemmArgs& g, f32x4 (&acc)[4][8], int wid, int lane, long m0, long n0,
                                                char* red) {
#pragma clang fp contract(off)
  constexpr int MT = 8;
  const int fr = lane & 15, fq = lane >> 4;
  const int wu = __builtin_amdgcn_readfirstlane(wid);
  const long nu = n0 + (wu & 3) * 64;
  long mu = m0 + (wu >> 2) * 128;
  const bool inside = mu < g.M;                       // M % 128 == 0 (entry): a wave's rows are all inside or all outside
  if (!inside) mu = g.M - 128;                        // (outside: same work on clamped addresses, no stores -- the barrier is for all)
  const int sec = nu >= g.qn_dmodel ? 1 : 0;
  const int hcol = (int)(nu - (long)sec * g.qn_dmodel);
  const int head = hcol >> 7, hb = hcol & 64;
  const uint32_t bu = (uint32_t)mu / (uint32_t)g.c.rpb;
  const int t0 = (int)((uint32_t)mu - bu * (uint32_t)g.c.rpb);
  const int f0 = lane_feat(fq, 0), f1 = lane_feat(fq, 2);
  uint4 bias0 = make_uint4(0, 0, 0, 0), bias1 = bias0;
  if (g.bias) {
    bias0 = *reinterpret_cast<const uint4*>(g.bias + nu + f0);
    bias1 = *reinterpret_cast<const uint4*>(g.bias + nu + f1);
  }
  // cos / sin rows of the lane: sequence position s0 + t0 + fr + 16 j, features hb + f0 .. +7 and hb + f1 .. +7
  const float* cbase = g.qn_cos + (long)(g.qn_s0 + t0 + fr) * 128 + hb;
  const float* sbase = g.qn_sin + (long)(g.qn_s0 + t0 + fr) * 128 + hb;
  // ring of table rows in flight ahead of the arithmetic.  The epilogue's cost IS these loads -- with the general tables 512 KiB
  // per tile through the CU's L1: 1.164 ms fused, 1.037 ms without them, 1.020 ms the plain projection, 0.212 ms the norm pass it
  // replaces (profiles/r04_qknorm_epilogue_prices.log).  PAIRED (EPI_QKNORM_P): a (cos, sin)-per-pair table for tables that repeat
  // every pair's entry (FLUX's) -- half the bytes, half the registers per row, so two rows ahead: 1.065 ms
  // (r04_qknorm_epilogue_prices_pair_table.log), once its arithmetic is the one-instruction helpers above (as plain C++ the SLP pass
  // packs it and the kernel spills 200 bytes per lane: 1.217 ms; three rows ahead spills again).
  constexpr int AHEAD = PAIRED ? 2 : 1, NSLOT = AHEAD + 1;      // the pair form's rows are half the registers: two rows ahead
  float4 cs[NSLOT][PAIRED ? 4 : 8];                   // [ring slot][cos f0, cos f0+4, cos f1, cos f1+4, sin ...] / [(cos, sin) x 4 pairs] x 4
  const float* pbase = g.qn_cs2 + (long)(g.qn_s0 + t0 + fr) * 128 + hb;      // (64 pairs x 2 floats = 128 floats per row as well)
  auto load_cs = [&](int slot, int j, uint32_t dep) {
    const long o = (long)(j * 16 + dep) * 128;
    if constexpr (PAIRED) {
      cs[slot][0] = *reinterpret_cast<const float4*>(pbase + o + f0);
      cs[slot][1] = *reinterpret_cast<const float4*>(pbase + o + f0 + 4);
      cs[slot][2] = *reinterpret_cast<const float4*>(pbase + o + f1);
      cs[slot][3] = *reinterpret_cast<const float4*>(pbase + o + f1 + 4);
      return;
    }
    cs[slot][0] = *reinterpret_cast<const float4*>(cbase + o + f0);
    cs[slot][1] = *reinterpret_cast<const float4*>(cbase + o + f0 + 4);
    cs[slot][2] = *reinterpret_cast<const float4*>(cbase + o + f1);
    cs[slot][3] = *reinterpret_cast<const float4*>(cbase + o + f1 + 4);
    cs[slot][4] = *reinterpret_cast<const float4*>(sbase + o + f0);
    cs[slot][5] = *reinterpret_cast<const float4*>(sbase + o + f0 + 4);
    cs[slot][6] = *reinterpret_cast<const float4*>(sbase + o + f1);
    cs[slot][7] = *reinterpret_cast<const float4*>(sbase + o + f1 + 4);
  };
  // y = bf16(acc + bias), the value the unfused path stores and reads back
  uint4 y[MT][2];
  {
    float bias[16];
    unpack8(bias0, bias);
    unpack8(bias1, bias + 8);
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      float v[16];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[t * 4 + r] = acc[t][j][r] + bias[t * 4 + r];
      y[j][0] = pack8(v);
      y[j][1] = pack8(v + 8);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // (only now, with the 128 accumulator registers retired into 64: the first table rows and the norm weights; the sums of
  //  squares and the exchange below cover their latency)
  // RMSNorm weights of the lane's 16 features (q: times q_scale, as mgx_qk_norm_rope_fwd_qs folds it)
  uint32_t dep0 = 0;                                  // an opaque zero that exists only once y is complete: pins the loads below
  asm volatile("" : "+v"(dep0) : "v"(y[MT - 1][1].w));   // behind the accumulators' retirement (they were hoisted above it, into spills)
  float wv[16];
  {
    const float* wp = (sec ? g.qn_wk : g.qn_wq) + hb + dep0;
    const float4 a0 = *reinterpret_cast<const float4*>(wp + f0), a1 = *reinterpret_cast<const float4*>(wp + f0 + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(wp + f1), b1 = *reinterpret_cast<const float4*>(wp + f1 + 4);
    const float t[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    const float qs = sec ? 1.0f : g.qn_qscale;
#pragma unroll
    for (int e = 0; e < 16; ++e) wv[e] = t[e] * qs;
  }
#pragma unroll
  for (int j = 0; j < AHEAD; ++j) load_cs(j, j, dep0);
  __builtin_amdgcn_sched_barrier(0);
  // sums of squares of the wave's half head, per row
  float ssq[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    float x[16], p[8];
    unpack8(y[j][0], x);
    unpack8(y[j][1], x + 8);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      p[i] = PAIRED ? vadd1(vmul1(x[2 * i], x[2 * i]), vmul1(x[2 * i + 1], x[2 * i + 1])) : x[2 * i] * x[2 * i] + x[2 * i + 1] * x[2 * i + 1];
    float sacc = PAIRED ? vadd1(vadd1(vadd1(p[0], p[1]), vadd1(p[2], p[3])), vadd1(vadd1(p[4], p[5]), vadd1(p[6], p[7])))
                        : ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));     // pair-index bits 0, 1, then 4
    sacc += __shfl_xor(sacc, 16, 64);                                                   // bit 2
    sacc += __shfl_xor(sacc, 32, 64);                                                   // bit 3
    ssq[j] = sacc;
    if (j & 1) __builtin_amdgcn_sched_barrier(0);     // two rows at a time: all eight interleaved unpack into 128 registers and spill
  }
  {
    float* wr = reinterpret_cast<float*>(red + (wu ^ 1) * 1024);
    const float* rd = reinterpret_cast<const float*>(red + wu * 1024);
    if (fq == 0) {
#pragma unroll
      for (int j = 0; j < MT; ++j) wr[fr + 16 * j] = ssq[j];
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < MT; ++j) ssq[j] = ssq[j] + rd[fr + 16 * j];                     // bit 5: the other half of the head
  }
  bf16_raw* out = (sec ? g.qn_K : g.qn_Q) + (((long)bu * g.qn_H + head) * g.qn_S + g.qn_s0 + t0 + fr) * 128 + hb;
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int slot = j % NSLOT;
    if (j + AHEAD < MT) {
      uint32_t dep = 0;
      if (j > 0) asm volatile("" : "+v"(dep) : "v"(y[j - 1][1].w));     // keeps the loads from being hoisted above row j - 1
      load_cs((j + AHEAD) % NSLOT, j + AHEAD, dep);
    }
    __builtin_amdgcn_sched_barrier(0);
    const float r = rsqrtf(ssq[j] / 128.f + 1e-6f);
    float x[16], o[16];
    unpack8(y[j][0], x);
    unpack8(y[j][1], x + 8);
    if constexpr (PAIRED) {
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {                 // (cos, sin) of the lane's pairs 2 q4 and 2 q4 + 1
        const float4 v = cs[slot][q4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int i = 2 * q4 + h;
          const float cv = h ? v.z : v.x, sv = h ? v.w : v.y;
          const float y0 = vmul1(vmul1(x[2 * i], r), wv[2 * i]), y1 = vmul1(vmul1(x[2 * i + 1], r), wv[2 * i + 1]);
          o[2 * i] = vsub1(vmul1(y0, cv), vmul1(y1, sv));
          o[2 * i + 1] = vadd1(vmul1(y1, cv), vmul1(y0, sv));
        }
      }
    } else {
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const float4 cv = cs[slot][q4], sv = cs[slot][4 + q4];
        const float c[4] = {cv.x, cv.y, cv.z, cv.w}, sn[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int i = 2 * q4 + h;
          const float y0 = x[2 * i] * r * wv[2 * i], y1 = x[2 * i + 1] * r * wv[2 * i + 1];
          o[2 * i] = y0 * c[2 * h] - y1 * sn[2 * h];
          o[2 * i + 1] = y1 * c[2 * h + 1] + y0 * sn[2 * h + 1];
        }
      }
    }
    y[j][0] = pack8(o);
    y[j][1] = pack8(o + 8);
    if (inside) {
      *reinterpret_cast<uint4*>(out + (long)j * 16 * 128 + f0) = y[j][0];
      *reinterpret_cast<uint4*>(out + (long)j * 16 * 128 + f1) = y[j][1];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int EPI>
__device__ __forceinline__ void persist_epilogue(const GemmArgs& g, f32x4 (&acc)[4][8], int wid, int lane, long m0,
                                                 long n0, char* red = nullptr) {
  if constexpr (EPI == EPI_QKNORM || EPI == EPI_QKNORM_P) {
    qknorm_epilogue<EPI == EPI_QKNORM_P>(g, acc, wid, lane, m0, n0, red);
    return;
  }
  constexpr int MT = 8;
  const int wm = wid >> 2, wn = wid & 3, fr = lane & 15, fq = lane >> 4;
  const long nw = n0 + wn * 64;                   // the wave's 64 output features; tile t of this lane: nw + lane_feat(fq, t)
  // N % 4 == 0 always; on the 16-byte paths N % 8 == 0: the two 8-feature halves are in range independently
  const long mrow0 = m0 + wm * 128 + fr;          // rows mrow0 + 16 j

  if (EPI == EPI_F32_ACC) {
    float* Cb = reinterpret_cast<float*>(g.C);
    const bool rmw = g.beta != 0.f;
    bool nok[4];
    long ncol[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      nok[t] = nw + lane_feat(fq, t) < g.N;
      ncol[t] = nok[t] ? nw + lane_feat(fq, t) : 0;
    }
#pragma unroll
    for (int jb = 0; jb < MT; jb += 2) {          // two rows (8 float4) of read-modify-write loads in flight
      long off[2];
      bool rok[2];
      float4 old[2][4];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const long m = mrow0 + (jb + jj) * 16;
        rok[jj] = m < g.M;
        off[jj] = rok[jj] ? row_off(g.c, m) : 0;
      }
      if (rmw) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int t = 0; t < 4; ++t) old[jj][t] = *reinterpret_cast<const float4*>(Cb + off[jj] + ncol[t]);
      }
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const f32x4 a = acc[t][jb + jj];
          float4 o = make_float4(a[0], a[1], a[2], a[3]);
          if (rmw) {                                  // one v_fma_f32 each: as v_pk_fma_f32 (what -O3's SLP pass makes of the four) this
            o.x = vfma1(g.beta, old[jj][t].x, o.x);   // epilogue ran 2 % slower beside the partner wave's MFMAs (profiles/r04_gemm_noslp_ab.log)
            o.y = vfma1(g.beta, old[jj][t].y, o.y);
            o.z = vfma1(g.beta, old[jj][t].z, o.z);
            o.w = vfma1(g.beta, old[jj][t].w, o.w);
          }
          if (rok[jj] && nok[t]) *reinterpret_cast<float4*>(Cb + off[jj] + ncol[t]) = o;
        }
    }
    return;
  }

  if (!g.rowwise_ok) {
    // generic (side operands only 8-byte addressable): 4 features per access
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const long m = mrow0 + j * 16;
      if (m >= g.M) continue;
      const long crow = row_off(g.c, m);
      const long bidx = (uint32_t)m / (uint32_t)g.c.rpb;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const long nn = nw + lane_feat(fq, t);
        if (nn >= g.N) continue;
        float v[4] = {acc[t][j][0], acc[t][j][1], acc[t][j][2], acc[t][j][3]};
        if (g.bias) {
          const uint2 bb = *reinterpret_cast<const uint2*>(g.bias + nn);
          v[0] += bf2f(bb.x & 0xffff); v[1] += bf2f(bb.x >> 16); v[2] += bf2f(bb.y & 0xffff); v[3] += bf2f(bb.y >> 16);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = rbf(v[r]);
        bf16_raw* cp = reinterpret_cast<bf16_raw*>(g.C) + crow + nn;
        uint2 pre;
        pre.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
        pre.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
        if (EPI == EPI_BIAS_GELU) {
          if (g.aux) *reinterpret_cast<uint2*>(g.aux + m * g.ldaux + nn) = pre;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_f(v[r]);
        } else if (EPI == EPI_BIAS_GATE_RES) {
          if (g.aux) *reinterpret_cast<uint2*>(g.aux + m * g.ldaux + nn) = pre;
          const uint2 gg = *reinterpret_cast<const uint2*>(g.gate + bidx * g.gate_ld + nn);
          const uint2 rr = *reinterpret_cast<const uint2*>(cp);
          v[0] = bf2f(rr.x & 0xffff) + rbf(bf2f(gg.x & 0xffff) * v[0]);
          v[1] = bf2f(rr.x >> 16) + rbf(bf2f(gg.x >> 16) * v[1]);
          v[2] = bf2f(rr.y & 0xffff) + rbf(bf2f(gg.y & 0xffff) * v[2]);
          v[3] = bf2f(rr.y >> 16) + rbf(bf2f(gg.y >> 16) * v[3]);
        } else if (EPI == EPI_BIAS_MULAUX) {
          const uint2 pp = *reinterpret_cast<const uint2*>(g.aux + m * g.ldaux + nn);
          v[0] *= gelu_tanh_grad_f(bf2f(pp.x & 0xffff)); v[1] *= gelu_tanh_grad_f(bf2f(pp.x >> 16));
          v[2] *= gelu_tanh_grad_f(bf2f(pp.y & 0xffff)); v[3] *= gelu_tanh_grad_f(bf2f(pp.y >> 16));
        }
        uint2 o;
        o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
        o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
        *reinterpret_cast<uint2*>(cp) = o;
      }
    }
    return;
  }

  // ---- 16-byte path.  Order: (1) request bias, gate and the side operands of rows 0..3; (2) retire all 128
  // accumulator registers into 64 registers of packed bf16 Linear outputs y = bf16(acc + bias); (3) finish the rows in
  // 2-row batches, each batch re-requesting the side operands four rows ahead BEFORE its own stores (vmcnt is in order
  // and counts stores: a load issued after a store cannot be waited for without draining the store).
  // Addressing: wave-uniform base (scalar registers) + one 32-bit per-lane byte offset per row; rows / columns outside
  // the matrix are clamped to the last valid row / the wave's first column for the loads and masked for the stores.
  // The wave's 128 rows lie in ONE batch (c_rpb % 128 == 0 or a plain matrix: part of rowwise_ok).
  const int wu = __builtin_amdgcn_readfirstlane(wid);
  const long mu = m0 + (wu >> 2) * 128, nu = n0 + (wu & 3) * 64;
  if (mu >= g.M || nu >= g.N) return;               // wave-uniform: nothing of this wave's block is inside the matrix
  const long bu = (uint32_t)mu / (uint32_t)g.c.rpb;
  long ncol = nu;                                   // the wave's first column inside C's row (column batches: mgx_linear_bf16_t)
  if (EPI == EPI_BIAS && g.col_rpb > 0) {
    const uint32_t cbt = (uint32_t)nu / (uint32_t)g.col_rpb;
    ncol = (long)cbt * g.col_bstride + (long)((uint32_t)nu - cbt * (uint32_t)g.col_rpb);
  }
  const char* cbase = reinterpret_cast<const char*>(g.C) + (bu * g.c.bstride + (mu - bu * g.c.rpb) * g.c.ld + ncol) * 2;
  const char* abase = reinterpret_cast<const char*>(g.aux) + (mu * g.ldaux + nu) * 2;
  const int rmax = (int)(g.M - 1 - mu < 127 ? g.M - 1 - mu : 127);      // last valid row of the block
  const int f0 = lane_feat(fq, 0), f1 = lane_feat(fq, 2);                // first feature of the lane's two 8-feature halves
  const bool nok0 = nu + f0 < g.N, nok1 = nu + f1 < g.N;                 // N % 8 == 0 on this path
  const uint32_t c0 = nok0 ? f0 * 2 : 0, c1 = nok1 ? f1 * 2 : 0;         // their byte offsets
  const uint32_t ldc2 = (uint32_t)(g.c.ld * 2), lda2 = (uint32_t)(g.ldaux * 2);
  auto rowclamp = [&](int j) { const int r = fr + 16 * j; return (uint32_t)(r < rmax ? r : rmax); };

  uint4 bias0 = make_uint4(0, 0, 0, 0), bias1 = bias0;
  const bool brows = EPI == EPI_BIAS && g.bias_rows && g.bias;      // wave-uniform
  float brow[MT];
  if (brows) {
#pragma unroll
    for (int j = 0; j < MT; ++j) brow[j] = bf2f(g.bias[mu + rowclamp(j)]);
  } else if (g.bias) {
    const char* bb = reinterpret_cast<const char*>(g.bias) + nu * 2;
    bias0 = *reinterpret_cast<const uint4*>(bb + c0);
    bias1 = *reinterpret_cast<const uint4*>(bb + c1);
  }
  constexpr bool SIDE = EPI == EPI_BIAS_GATE_RES || EPI == EPI_BIAS_MULAUX;
  uint4 gate0 = make_uint4(0, 0, 0, 0), gate1 = gate0;
  if (EPI == EPI_BIAS_GATE_RES) {
    const char* gp = reinterpret_cast<const char*>(g.gate) + (bu * g.gate_ld + nu) * 2;
    gate0 = *reinterpret_cast<const uint4*>(gp + c0);
    gate1 = *reinterpret_cast<const uint4*>(gp + c1);
  }
  uint4 side[2][SIDE ? 2 : 1][2];                   // [ring slot][row of the 2-row batch][feature half]
  // `dep` is an opaque zero that is data-dependent on the previous batch's results: without it hipcc hoists ALL side
  // loads to the top of the epilogue, runs out of registers and spills the loaded data (scratch + vmcnt(0) per load)
  auto load_side = [&](int h, int j0, uint32_t dep) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const uint32_t ro = (rowclamp(j0 + k) + dep) * (EPI == EPI_BIAS_GATE_RES ? ldc2 : lda2);
      const char* base = EPI == EPI_BIAS_GATE_RES ? cbase : abase;
      side[h][SIDE ? k : 0][0] = *reinterpret_cast<const uint4*>(base + (ro + c0));
      side[h][SIDE ? k : 0][1] = *reinterpret_cast<const uint4*>(base + (ro + c1));
    }
  };
  if (SIDE) {
    load_side(0, 0, 0);
    load_side(1, 2, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  uint4 y[MT][2];
  {
    float bias[16];
    unpack8(bias0, bias);
    unpack8(bias1, bias + 8);
    if (brows) {                                     // (a wave-uniform branch around the loop: no per-element select)
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        float v[16];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[t * 4 + r] = acc[t][j][r] + brow[j];
        y[j][0] = pack8(v);
        y[j][1] = pack8(v + 8);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        float v[16];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[t * 4 + r] = acc[t][j][r] + bias[t * 4 + r];
        y[j][0] = pack8(v);
        y[j][1] = pack8(v + 8);
        __builtin_amdgcn_sched_barrier(0);           // row by row: interleaving the rows doubles the live registers
      }
    }
  }
#pragma unroll
  for (int jb = 0; jb < MT; jb += 2) {
    const int slot = (jb >> 1) & 1;
    uint4 o[2][2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = jb + jj;
      if (EPI == EPI_BIAS) {
        o[jj][0] = y[j][0];
        o[jj][1] = y[j][1];
      } else {
        float v[16];
        unpack8(y[j][0], v);
        unpack8(y[j][1], v + 8);
        if (EPI == EPI_BIAS_GELU) {
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] = gelu_tanh_f(v[e]);
        } else if (EPI == EPI_BIAS_GATE_RES) {
          float gv[16], rv[16];
          unpack8(gate0, gv);
          unpack8(gate1, gv + 8);
          unpack8(side[slot][SIDE ? jj : 0][0], rv);
          unpack8(side[slot][SIDE ? jj : 0][1], rv + 8);
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] = rv[e] + rbf(gv[e] * v[e]);
        } else if (EPI == EPI_BIAS_MULAUX) {
          float pv[16];
          unpack8(side[slot][SIDE ? jj : 0][0], pv);
          unpack8(side[slot][SIDE ? jj : 0][1], pv + 8);
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] *= gelu_tanh_grad_f(pv[e]);
        }
        o[jj][0] = pack8(v);
        o[jj][1] = pack8(v + 8);
      }
    }
    if (SIDE && jb + 4 < MT) {                      // rows jb + 4, jb + 5 into the slot just consumed
      uint32_t dep = 0;
      asm volatile("" : "+v"(dep) : "v"(o[1][1].w));
      load_side(slot, jb + 4, dep);
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = jb + jj;
      const int r = fr + 16 * j;
      if (r <= rmax) {
        char* cb = const_cast<char*>(cbase);
        const uint32_t co = (uint32_t)r * ldc2;
        if ((EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_GATE_RES) && g.aux) {
          char* ab = const_cast<char*>(abase);      // pre-activation / pre-gate branch output for the backward pass
          const uint32_t ao = (uint32_t)r * lda2;
          if (nok0) *reinterpret_cast<uint4*>(ab + (ao + c0)) = y[j][0];
          if (nok1) *reinterpret_cast<uint4*>(ab + (ao + c1)) = y[j][1];
        }
        if (nok0) *reinterpret_cast<uint4*>(cb + (co + c0)) = o[jj][0];
        if (nok1) *reinterpret_cast<uint4*>(cb + (co + c1)) = o[jj][1];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}


// ------------------------------------------------------------------------------------------ stream-K tail
// A persistent launch takes ceil(tiles / 256) rounds, and the step is full of shapes whose last round is mostly idle: 2.25
// rounds (the 3072 x 12288 weight gradients), 0.56 (3072 x 3072), 0.375 / 0.75 (the text stream), 3.375 / 4.2 (N = 3072 at
// micro-batch 4 / 5): 0.55 s of a 13.1 s GEMM family (profiles/r03_gemm_shapes.jsonl).  So the LAST, partial round of every XCD
// is shared out along K, IN LOCKSTEP: with R tiles left for the XCD's nw workgroups, every tile's K range is cut into
// P = nw / R equal parts (2 <= P <= SK_PMAX) and workgroup w = p R + r runs part p of tile r -- the R workgroups of a part walk
// the same K-tiles at the same time, so they share their A / W panels in the XCD's L2 exactly as the workgroups of a whole
// round do (a first version dealt contiguous runs of tile slices, every workgroup at its own K offset: nothing was shared, the
// tail ran at the fabric's bandwidth and returned a third of what the K-loop arithmetic promised; profiles/r04_gemm_sk_ab_contiguous_runs.log vs r04_gemm_sk_ab_lockstep.log).
// Every part leaves its raw fp32 accumulators in the workspace (slot = blockIdx) and gemm_sk_fixup_kernel adds a tile's parts
// IN K ORDER and runs the ordinary epilogue: deterministic (no atomics, no flags, no spinning), and a given shape is always
// split the same way, so the training forward and its recompute stay bit-identical.  What a split buys is (1 - 1 / P) of a
// tile's K-loop, what it costs is the workspace round trip and the fix-up launch; launch() only allows it when the former is
// clearly larger (sk_minparts).
constexpr int SK_PMAX = 8;
constexpr long SK_SLOT = 256L * 256;      // floats per workspace slot

__host__ __device__ inline int sk_parts(int R, int nw) { return R > 0 ? (nw / R < SK_PMAX ? nw / R : SK_PMAX) : 0; }
__host__ __device__ inline bool sk_on(const float* ws, int minparts, int R, int nw, int nkt) {
  const int P = sk_parts(R, nw);
  return ws != nullptr && P >= 2 && P >= minparts && nkt >= 4 * P;
}

struct SkTail {
  int nfull;               // whole tiles of this workgroup (rounds before the tail)
  int nseg;                // tail units: 0 or 1
  int tile0, k00, k01, part0;
  int tile1, k11;          // (a second tail unit: unused by the lockstep dealing, kept for the unit walk's generality)
};

__device__ __forceinline__ SkTail sk_tail(const GemmArgs& g, int xcnt, int nw, int w, int nkt) {
  SkTail t;
  t.nfull = xcnt / nw;
  const int R = xcnt - t.nfull * nw, base = t.nfull * nw;
  t.nseg = 0; t.tile0 = 0; t.k00 = 0; t.k01 = nkt; t.part0 = 0; t.tile1 = 0; t.k11 = nkt;
  if (!sk_on(g.sk_ws, g.sk_minparts, R, nw, nkt)) {
    if (w < R) { t.nseg = 1; t.tile0 = base + w; }
    return t;
  }
  const int P = sk_parts(R, nw);
  if (w >= P * R) return t;
  const int p = w / R, r = w - p * R;
  t.nseg = 1; t.tile0 = base + r;
  t.k00 = p * nkt / P; t.k01 = (p + 1) * nkt / P;
  t.part0 = 1;
  return t;
}

// One wave per (tail tile, wave slot of the main kernel): blockIdx.x = (r * 8 + xcd) * 8 + wid.  Mirrors gemm_pp_kernel's
// tile walk and sk_tail()'s dealing.
template <int EPI, bool PAIR>
__global__ void __launch_bounds__(64) gemm_sk_fixup_kernel(GemmArgs g, int nw) {
  const int lane = threadIdx.x, wid = blockIdx.x & 7, xcd = (blockIdx.x >> 3) & 7, r = blockIdx.x >> 6;
  const int tiles_m = (g.M + 255) / 256, tiles_n = (g.N + 255) / 256, nwg = tiles_m * tiles_n, nkt = g.K / BK;
  const int q = nwg >> 3, rem = nwg & 7;
  const int xbeg = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
  const int xcnt = xcd < rem ? q + 1 : q;
  const int nfull = xcnt / nw, R = xcnt - nfull * nw;
  if (r >= R || !sk_on(g.sk_ws, g.sk_minparts, R, nw, nkt)) return;
  const int P = sk_parts(R, nw);
  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int p = 0; p < P; ++p) {                    // ascending p = ascending K
    const int w = p * R + r;
    const float* wsp = g.sk_ws + ((long)(w * 8 + xcd) << 16) + ((wid * 32) * 64 + lane) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(wsp + (i * 8 + j) * 256);
        acc[i][j] += v;
      }
  }
  const int tl = xbeg + nfull * nw + r, band = g.band;
  const int per_band = band * tiles_n, b0 = tl / per_band, rows_in_band = min(band, tiles_m - b0 * band);
  const int in_band = tl - b0 * per_band;
  long m0 = (long)(b0 * band + in_band % rows_in_band) * 256;
  const long n0 = (long)(in_band / rows_in_band) * 256;
  const GemmArgs gl = seg_args<PAIR>(g, m0);
  if constexpr (EPI != EPI_QKNORM && EPI != EPI_QKNORM_P) persist_epilogue<EPI>(gl, acc, wid, lane, m0, n0);   // (QKNORM launches never split: launch())
}

// ------------------------------------------------------------------------------------------ persistent ping-pong kernel
// 256 workgroups (one per CU) x 8 waves walk 256x256x64 tiles.  Staging: all 160 KiB of LDS as THREE 32 KiB stages for A and
// TWO for W, filled by LDS-DMA (global_load_lds_dwordx4, XOR-swizzled through the per-lane SOURCE address) two K-tiles ahead;
// raw s_barrier + counted vmcnt (__syncthreads() would drain the DMA queue); the pipeline runs on across output tiles.
// Tile order: bands of `band` tile rows, column-major inside a band, so the 32 CUs of an XCD work on band x (32 / band) tiles
// that share `band` A panels and 32 / band W panels.  The activation operand streams from HBM, the weights sit in the
// Infinity Cache: narrow outputs (<= 16 tile columns) take band 1 -- a round is whole tile rows, every A panel is fetched
// once -- wider ones band 4; few tile ROWS (wgrad shapes with 3072 output rows): one band of all rows, every W panel once.
// (Round 1's K-loop -- all eight waves in the same phase -- is kept as text in scratch/gemm_persist_kernel_round1.hip.txt.)
// The K-loop: a K-tile is TWO phases of 32 MFMAs per wave, each phase = [load section | barrier | MFMA section | barrier], and the second half of the waves
// (wm = 1: the SIMD partners of the first half) runs ONE barrier behind, so that on every SIMD one wave's MFMA section
// coincides with its partner's load section (LDS fragment reads + four LDS-DMA pieces): the in-order stalls of DMA issue and
// fragment reads never hold up a wave's own MFMAs.
//   LA: W fragments (8) + A fragments of token half 0 (8); A pieces 0-3 of K-tile k+2
//   MA: tokens 0-63 x features 0-63        LB: A fragments of token half 1 (8); W pieces 0-3 of K-tile k+2; vmcnt(8)
//   MB: tokens 64-127 x features 0-63
// LDS hazards by construction: A stage k+2 replaces stage k-1 (3 stages; last read in LB of k-1 -- of the late half during the
// early half's MB of k-1 -- one barrier before the first LA of k); W stage k+2 replaces stage k (2 stages; last read in LA of k,
// of the late half during the early half's MA of k, one barrier before the first LB of k).  vmcnt(8) at the end of LB leaves
// this K-tile's eight pieces in flight: everything K-tile k+1 reads has landed in every wave before the barrier in front of
// the first LA of k+1.
// Measured (profiles/r02_pp_clock.log, in-kernel clock = d s_memtime / d s_memrealtime): 2423-2476 shader cycles per K-tile
// against the matrix pipe's 2048 (83-85 % busy) at a clock the chip holds at 1.71-1.76 GHz under this load; the first
// version of this loop with FOUR phases of 16 MFMAs (8 barriers per K-tile) ran 2622-2703 cycles at 1.82-1.87 GHz and
// 2.5-6 % fewer TFLOP/s (profiles/r02_gemm_pp4_ab.log), round 1's lockstep K-loop 6-8 % fewer (profiles/r02_gemm_pp_ab.log).
template <int EPI, bool CONV, int MODE>
__global__ void __launch_bounds__(512, 2) gemm_pp_kernel(GemmArgs g) {
  constexpr bool SK = (MODE & 1) != 0, PAIR = (MODE & 2) != 0;
  constexpr int TM = 256, TN = 256, NTHR = 512, RS = NTHR / 8, TB = TM * 128, MT = 8, NTL = 4;
  constexpr int WBASE = 3 * TB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 2, wn = wid & 3;
  const int fr = lane & 15, fq = lane >> 4;
  const int tiles_m = (g.M + TM - 1) / TM, tiles_n = (g.N + TN - 1) / TN;
  const int nwg = tiles_m * tiles_n;
  const int nkt = g.K / BK;
  const int band = g.band;                // = tiles_n <= 16 ? 1 : (tiles_m <= 16 ? tiles_m : 4), launch()
  const int xcd = blockIdx.x & 7, lane_in_xcd = blockIdx.x >> 3, per_xcd_wg = gridDim.x >> 3;
  const int q = nwg >> 3, rem = nwg & 7;
  const int xbeg = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
  const int xcnt = xcd < rem ? q + 1 : q;
  // ---- the workgroup's list of work units: `nfull` whole tiles (xbeg + lane_in_xcd + i * per_xcd_wg), then -- stream-K tail --
  // up to two SEGMENTS (tile, K-tile range) of the XCD's last, partial round (SkTail)
  // (SK = false, every launch that splits nothing: the list is the strided tile walk and all of this folds away -- the unit
  //  bookkeeping costs the big rollout shapes ~1 %, profiles/r04_gemm_ab_r03_r04_a.log, so they do not carry it)
  SkTail sk;
  if (SK) {
    sk = sk_tail(g, xcnt, per_xcd_wg, lane_in_xcd, nkt);
  } else {
    sk.nfull = lane_in_xcd < xcnt ? (xcnt - lane_in_xcd + per_xcd_wg - 1) / per_xcd_wg : 0;
    sk.nseg = 0; sk.tile0 = 0; sk.k00 = 0; sk.k01 = nkt; sk.part0 = 0; sk.tile1 = 0; sk.k11 = nkt;
  }
  const int nunits = sk.nfull + sk.nseg;
  if (nunits == 0) return;
  const int wu = __builtin_amdgcn_readfirstlane(wid);
  const bool late = wu >= 4;
  const int lrow = tid >> 3, lkc = tid & 7;
  const int src_kc = swz(lrow, lkc);
  const int wrow = wperm(lrow);
  typedef __attribute__((address_space(3))) char lds_char;
  typedef const __attribute__((address_space(1))) char gbl_char;
  uint32_t ao[4], wo[4];
  long m0, n0;
#define TILE_COORDS(tl, M0, N0)                                            \
  do {                                                                     \
    const int per_band = band * tiles_n;                                   \
    const int b0 = (tl) / per_band;                                        \
    const int rows_in_band = min(band, tiles_m - b0 * band);               \
    const int in_band = (tl) - b0 * per_band;                              \
    M0 = (long)(b0 * band + in_band % rows_in_band) * TM;                  \
    N0 = (long)(in_band / rows_in_band) * TN;                              \
  } while (0)
#define TILE_OFFS(M0, N0, AO, WO)                                                          \
  do {                                                                                     \
    _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) {                                     \
      long mm = M0 + lrow + k_ * RS;                                                       \
      if (mm >= g.M) mm = g.M - 1;                                                         \
      const bool s2_ = PAIR && g.m_split && M0 >= g.m_split;                               \
      const long ar_ = !PAIR ? row_off(g.a, mm)                                            \
                             : (s2_ ? row_off(g.a2, mm - g.m_split) + g.a2_off : row_off(g.a, mm) + g.a1_off); \
      AO[k_] = (uint32_t)((ar_ + src_kc * 8) * 2);                                         \
      long nn = N0 + k_ * 64 + wrow;                                                       \
      if (nn >= g.N) nn = g.N - 1;                                                         \
      WO[k_] = (uint32_t)((nn * g.ldw + (!PAIR ? 0 : (s2_ ? g.w2_off : g.w1_off)) + src_kc * 8) * 2); \
    }                                                                                      \
  } while (0)
#define PGLDS(base, off, off_lds) \
  __builtin_amdgcn_global_load_lds((gbl_char*)((base) + (off)), (lds_char*)(smem + (off_lds)), 16, 0, 0)
#define PIN() __builtin_amdgcn_sched_barrier(0)
#define BAR() do { PIN(); __builtin_amdgcn_s_barrier(); PIN(); } while (0)

  uint32_t a_ro[2], w_ro[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_ro[ks] = (uint32_t)((wm * 128 + fr) * 128 + (swz(fr, ks * 4 + fq) << 4));
    w_ro[ks] = (uint32_t)(WBASE + (wn * 64 + fr) * 128 + (swz(fr, ks * 4 + fq) << 4));
  }
#define LDA(ks, j) (*reinterpret_cast<const s16x8*>(smem + sa_ + a_ro[ks] + (j) * 2048))
#define LDW(ks, i) (*reinterpret_cast<const s16x8*>(smem + sw_ + w_ro[ks] + (i) * 2048))
#define MMA(i_, j_, W_, A_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W_, A_, acc[i_][j_], 0, 0, 0)

#define PGLDS_K PGLDS
#define LDA_K LDA
#define LDW_K LDW
  // unit i of this workgroup -> (tile index inside the XCD's range, K-tile range, partial?)
#define UNIT(i, T_, K0_, K1_, P_)                                                        \
  do {                                                                                     \
    const int s_ = SK ? (i) - sk.nfull : -1;                                               \
    T_ = s_ < 0 ? lane_in_xcd + (i) * per_xcd_wg : (s_ == 0 ? sk.tile0 : sk.tile1);       \
    K0_ = s_ < 0 ? 0 : (s_ == 0 ? sk.k00 : 0);                                             \
    K1_ = s_ < 0 ? nkt : (s_ == 0 ? sk.k01 : sk.k11);                                      \
    P_ = s_ < 0 ? 0 : (s_ == 0 ? sk.part0 : 1);                                            \
  } while (0)
  int ui = 0, u_tile, kbeg, kend, partial;
  UNIT(0, u_tile, kbeg, kend, partial);
  TILE_COORDS(xbeg + u_tile, m0, n0);
  TILE_OFFS(m0, n0, ao, wo);
  {  // prologue: A K-tiles kbeg, kbeg + 1 -> A slots 0, 1; W likewise -> W slots 0, 1
    const char* ab = reinterpret_cast<const char*>(g.A);
    const char* wb = reinterpret_cast<const char*>(g.W) + (long)kbeg * (BK * 2);
    const int la = wu * 1024, lw = WBASE + wu * 1024;
#pragma unroll
    for (int k = 0; k < 4; ++k) PGLDS(ab + a_koff<CONV>(g, kbeg) * 2, ao[k], la + k * RS * 128);
#pragma unroll
    for (int k = 0; k < 4; ++k) PGLDS(wb, wo[k], lw + k * RS * 128);
#pragma unroll
    for (int k = 0; k < 4; ++k) PGLDS(ab + a_koff<CONV>(g, kbeg + 1) * 2, ao[k], TB + la + k * RS * 128);
#pragma unroll
    for (int k = 0; k < 4; ++k) PGLDS(wb + BK * 2, wo[k], TB + lw + k * RS * 128);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (late) __builtin_amdgcn_s_barrier();          // the late half runs one barrier behind from here on
  int aslot = 0, wslot = 0;
  while (true) {
    const bool has_next = ui + 1 < nunits;
    long nm0 = 0, nn0 = 0;
    int n_tile = u_tile, nkbeg = kbeg, nkend = kend, npartial = partial;   // no next unit: the look-ahead re-reads this one
    uint32_t nao[4] = {ao[0], ao[1], ao[2], ao[3]}, nwo[4] = {wo[0], wo[1], wo[2], wo[3]};
    if (has_next) {
      UNIT(ui + 1, n_tile, nkbeg, nkend, npartial);
      TILE_COORDS(xbeg + n_tile, nm0, nn0);
      TILE_OFFS(nm0, nn0, nao, nwo);
    }
    uint32_t ca[4] = {ao[0], ao[1], ao[2], ao[3]}, cw[4] = {wo[0], wo[1], wo[2], wo[3]};
    f32x4 acc[NTL][MT];
#pragma unroll
    for (int i = 0; i < NTL; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int klen = kend - kbeg;                   // >= 2 (sk_tail)
    for (int kt = 0; kt < klen; ++kt) {
      const bool nxt = kt + 2 >= klen;
      const int k2 = nxt ? nkbeg + (kt + 2 - klen) : kbeg + kt + 2;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ca[k] = nxt ? nao[k] : ca[k];
        cw[k] = nxt ? nwo[k] : cw[k];
      }
      const int a_dst = aslot == 0 ? 2 : aslot - 1;
      const char* ab_ = reinterpret_cast<const char*>(g.A) + a_koff<CONV>(g, k2) * 2;
      const char* wb_ = reinterpret_cast<const char*>(g.W) + (long)k2 * (BK * 2);
      const int la_ = a_dst * TB + wu * 1024, lw_ = WBASE + wslot * TB + wu * 1024;
      const uint32_t sa_ = aslot * TB, sw_ = wslot * TB;
      s16x8 faA[2][4], faB[2][4], fw[2][4];
      // ---- LA
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i) fw[ks][i] = LDW_K(ks, i);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) faA[ks][j] = LDA_K(ks, j);
#pragma unroll
      for (int k = 0; k < 4; ++k) PGLDS_K(ab_, ca[k], la_ + k * RS * 128);
      BAR();
      // ---- MA: tokens 0-63 x features 0-63
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) MMA(i, j, fw[ks][i], faA[ks][j]);
      __builtin_amdgcn_s_setprio(0);
      BAR();
      // ---- LB
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) faB[ks][j] = LDA_K(ks, 4 + j);
#pragma unroll
      for (int k = 0; k < 4; ++k) PGLDS_K(wb_, cw[k], lw_ + k * RS * 128);
      PIN();
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      BAR();
      // ---- MB: tokens 64-127 x features 0-63
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) MMA(i, 4 + j, fw[ks][i], faB[ks][j]);
      __builtin_amdgcn_s_setprio(0);
      BAR();
      aslot = aslot == 2 ? 0 : aslot + 1;
      wslot ^= 1;
    }
    int el = lane, ew = wid;
    asm volatile("" : "+v"(el), "+v"(ew));
    // (measured on the four-phase version and dropped: both halves running their epilogues in the SAME barrier interval --
    //  one extra barrier per half and tile -- is neutral; `s_setprio` around the MFMA sections is worth 2 %)
    if (SK && partial) {
      // stream-K part: the raw accumulators go to this workgroup's workspace slot in register order (every store
      // instruction of a wave writes 1 KiB contiguous); gemm_sk_fixup_kernel sums a tile's parts and runs the epilogue
      float* wsp = g.sk_ws + ((long)blockIdx.x << 16) + ((ew * 32) * 64 + el) * 4;
#pragma unroll
      for (int i = 0; i < NTL; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) *reinterpret_cast<f32x4*>(wsp + (i * MT + j) * 256) = acc[i][j];
    } else if (PAIR) {
      long m0l = m0;
      const GemmArgs gl = seg_args<true>(g, m0l);
      persist_epilogue<EPI>(gl, acc, ew, el, m0l, n0);
    } else {
      // (EPI_QKNORM: the A stage the next K-tile's DMA will fill -- its last readers, the late half's LB of this tile's last
      //  K-tile, are one barrier back for the early half and two for the late one)
      persist_epilogue<EPI>(g, acc, ew, el, m0, n0, smem + (aslot == 0 ? 2 : aslot - 1) * TB);
    }
    if (!has_next) break;
    ++ui;
    u_tile = n_tile; kbeg = nkbeg; kend = nkend; partial = npartial;
    m0 = nm0; n0 = nn0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { ao[k] = nao[k]; wo[k] = nwo[k]; }
  }
  if (!late) __builtin_amdgcn_s_barrier();         // matches the late half's last barrier
#undef UNIT
#undef TILE_COORDS
#undef TILE_OFFS
#undef PGLDS
#undef PIN
#undef BAR
#undef PGLDS_K
#undef LDA_K
#undef LDW_K
#undef LDA
#undef LDW
#undef MMA
}

// what the persistent 256x256 kernel takes (the rest goes to gemm_kernel): enough tiles to fill most of the 256 CUs
// (measured, scratch/bench_gemm_small.py: 168 tiles 843-935 vs 591-691 TFLOP/s, 120 tiles equal, 24-96 tiles slower)
bool persistent_ok(const GemmArgs& g) {
  const long tiles_big = (long)cdiv(g.M, 256) * cdiv(g.N, 256);
  static const long min_tiles = getenv("MGX_GEMM_BIG_MIN_TILES") ? atol(getenv("MGX_GEMM_BIG_MIN_TILES")) : 128;
  static const int mode = getenv("MGX_GEMM_MODE") ? atoi(getenv("MGX_GEMM_MODE")) : 9;     // 0 (debugging): 128x128 kernel everywhere
  const bool big = g.M >= 256 && g.N >= 256 && tiles_big >= min_tiles && (g.N % 256 == 0 || g.N >= 2048);
  return big && mode != 0 && g.span32 && g.K >= 2 * BK;
}

template <int EPI, bool CONV = false>
int launch(const GemmArgs& g_in, hipStream_t st) {
  GemmArgs g = g_in;
  {
    const int tiles_m = cdiv(g.M, 256), tiles_n = cdiv(g.N, 256);
    static const int band_env = getenv("MGX_GEMM_BAND") ? atoi(getenv("MGX_GEMM_BAND")) : 0;      // A/B only
    g.band = band_env > 0 ? min(band_env, tiles_m) : (g.band > 0 ? min(g.band, tiles_m) : (tiles_n <= 16 ? 1 : (tiles_m <= 16 ? tiles_m : 4)));
  }
  const long tiles_big = (long)cdiv(g.M, 256) * cdiv(g.N, 256);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_kernel<EPI, CONV>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<EPI, CONV, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    if (!CONV) {
      (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<EPI, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
      (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<EPI, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
      (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<EPI, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    }
    attr_set = true;
  }
  const bool pair = !CONV && g.m_split != 0;
  if (EPI == EPI_QKNORM || EPI == EPI_QKNORM_P) g.sk_ws = nullptr;     // its epilogue needs the whole workgroup: no split tiles
  if (persistent_ok(g)) {
    int grid = 256;                       // one workgroup per CU (multiple of 8: XCD ranges)
    // stream-K tail (caller gave a workspace): allowed when it shortens the launch.  A tile's K-loop takes T ~ K * 0.0247 us
    // (2 * 256 * 256 * K FLOP at the 5.3 TFLOP/s a CU sustains in this kernel).  Unsplit, the last round takes T; split P ways
    // it takes T / P -- times a slowdown for the P-fold panel traffic of the tail (P parts x (rows + columns) panels per step
    // against one round's) -- plus ~40 us for the workspace round trip and the fix-up launch:
    //     split  <=>  T * (1 - slow / P) > cost.
    static const float sk_cost_us = getenv("MGX_GEMM_SK_COST_US") ? (float)atof(getenv("MGX_GEMM_SK_COST_US")) : 40.f;
    static const float sk_slow = getenv("MGX_GEMM_SK_SLOWDOWN") ? (float)atof(getenv("MGX_GEMM_SK_SLOWDOWN")) : 1.25f;
    static const int sk_off = getenv("MGX_GEMM_SK") ? atoi(getenv("MGX_GEMM_SK")) == 0 : 0;
    int rmax = 0;
    g.sk_minparts = 0;
    if (g.sk_ws && !sk_off && !CONV) {
      const float tile_us = (float)g.K * 0.0247f;
      const float room = 1.f - sk_cost_us / tile_us;                 // split <=> slow / P < room
      g.sk_minparts = room <= 0.f ? SK_PMAX + 1 : (int)floorf(sk_slow / room) + 1;
      if (g.sk_minparts < 2) g.sk_minparts = 2;
      const int nw = grid / 8, q = (int)(tiles_big >> 3), rem = (int)(tiles_big & 7);
      for (int x = 0; x < 8; ++x) {
        const int cnt = x < rem ? q + 1 : q, R = cnt % nw;
        if (sk_on(g.sk_ws, g.sk_minparts, R, nw, g.K / BK) && R > rmax) rmax = R;
      }
    }
    if (rmax == 0) {
      g.sk_ws = nullptr;
      if (tiles_big < grid) grid = (int)((tiles_big + 7) / 8 * 8);
    }
    if (rmax > 0 && pair) {
      gemm_pp_kernel<EPI, false, 3><<<grid, 512, 163840, st>>>(g);
      gemm_sk_fixup_kernel<EPI, true><<<rmax * 64, 64, 0, st>>>(g, grid / 8);
    } else if (rmax > 0) {
      gemm_pp_kernel<EPI, false, 1><<<grid, 512, 163840, st>>>(g);
      gemm_sk_fixup_kernel<EPI, false><<<rmax * 64, 64, 0, st>>>(g, grid / 8);
    } else if (pair) {
      gemm_pp_kernel<EPI, false, 2><<<grid, 512, 163840, st>>>(g);
    } else {
      gemm_pp_kernel<EPI, CONV, 0><<<grid, 512, 163840, st>>>(g);
    }
  } else {
    if (pair) return 1;                   // only the persistent kernel walks two problems: the caller launches them one by one
    gemm_kernel<EPI, CONV><<<cdiv(g.M, BM) * cdiv(g.N, BN), NT, 65536, st>>>(g);
  }
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

// y = bf16(gelu_tanh(float(x))) row by row: exactly what the bias+GELU epilogue writes as C from the pre-activation it
// writes as aux (the epilogue applies GELU to the bf16-ROUNDED Linear output), so a training pass that KEPT the pre-activation
// re-creates the activation without re-running the GEMM (mixgrpo_amd/flux_backward.py, `_Train.keep[...]["hid_pre"]`;
// tests/test_hip_gemm.py holds the two to bit-identity on every bf16 value).
__global__ void __launch_bounds__(256) gelu_rows_kernel(const bf16_raw* __restrict__ x, long ldx, bf16_raw* __restrict__ y,
                                                        long ldy, long M, int N8) {
  const long total = M * N8;
  for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
    const long m = id / N8;
    const int c = (int)(id - m * N8);
    const uint4 u = *reinterpret_cast<const uint4*>(x + m * ldx + c * 8);
    float v[8];
    unpack8(u, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = gelu_tanh_f(v[e]);
    *reinterpret_cast<uint4*>(y + m * ldy + c * 8) = pack8(v);
  }
}

}  // namespace

extern "C" long mgx_gemm_sk_workspace_elems(void) { return 256L * SK_SLOT; }

namespace {
int dispatch(const GemmArgs& g, int epilogue, hipStream_t st) {
  switch (epilogue) {
    case EPI_BIAS: return launch<EPI_BIAS>(g, st);
    case EPI_BIAS_GELU: return launch<EPI_BIAS_GELU>(g, st);
    case EPI_BIAS_GATE_RES: return launch<EPI_BIAS_GATE_RES>(g, st);
    case EPI_F32_ACC: return launch<EPI_F32_ACC>(g, st);
    case EPI_BIAS_MULAUX: return launch<EPI_BIAS_MULAUX>(g, st);
  }
  mgx_set_error("unknown GEMM epilogue");
  return MGX_ERR_ARG;
}

bool side_ok16(int N, int M, const void* C, long ldc, long c_rpb, long c_bstride, const void* aux, long ldaux, const void* gate,
               long gate_ld, const void* bias) {
  return (N % 8 == 0) && (c_rpb >= M || c_rpb % 128 == 0) && (ldc % 8 == 0) && (c_bstride % 8 == 0) && ((uintptr_t)C % 16 == 0) &&
         (!aux || (ldaux % 8 == 0 && (uintptr_t)aux % 16 == 0)) && (!gate || (gate_ld % 8 == 0 && (uintptr_t)gate % 16 == 0)) &&
         (!bias || (uintptr_t)bias % 16 == 0);
}
}  // namespace

extern "C" int mgx_gemm_bf16_sk(const uint16_t* A, const uint16_t* W, const uint16_t* bias, void* C, const uint16_t* gate,
                                uint16_t* aux, long ldaux, int M, int N, int K, long lda, long a_rpb, long a_bstride, long ldw, long ldc,
                                long c_rpb, long c_bstride, long gate_ld, int epilogue, float beta, float* sk_workspace,
                                long sk_workspace_elems, void* stream) {
  MGX_REQUIRE(A && W && C, "null operand");
  MGX_REQUIRE(!sk_workspace || (sk_workspace_elems >= mgx_gemm_sk_workspace_elems() && (uintptr_t)sk_workspace % 16 == 0),
              "stream-K workspace too small (mgx_gemm_sk_workspace_elems) or misaligned");
  MGX_REQUIRE(M > 0 && N > 0 && K > 0, "empty GEMM");
  MGX_REQUIRE(K % BK == 0, "K must be a multiple of 64");
  MGX_REQUIRE(N % 4 == 0, "N must be a multiple of 4");
  MGX_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0, "leading dimensions must keep 16-byte row alignment");
  MGX_REQUIRE(a_rpb > 0 && c_rpb > 0, "rows-per-batch must be positive");
  MGX_REQUIRE(!aux || ldaux % 4 == 0, "aux leading dimension must keep 8-byte alignment");
  MGX_REQUIRE(a_bstride % 8 == 0 && c_bstride % 4 == 0, "batch strides must keep alignment");
  MGX_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0) && ((uintptr_t)C % 8 == 0), "operands must be 16-byte aligned");
  MGX_REQUIRE(epilogue != EPI_BIAS_GATE_RES || gate, "gate-residual epilogue needs a gate");
  MGX_REQUIRE(epilogue != EPI_BIAS_MULAUX || aux, "gelu-backward epilogue needs the saved pre-activation");
  GemmArgs g{};
  g.A = A; g.W = W; g.bias = bias; g.C = C; g.gate = gate; g.aux = aux; g.ldaux = ldaux; g.gate_ld = gate_ld;
  g.M = M; g.N = N; g.K = K;
  const long RPB_MAX = 1L << 30;
  g.a = RowMap{lda, a_rpb < RPB_MAX ? a_rpb : RPB_MAX, a_bstride};
  g.c = RowMap{ldc, c_rpb < RPB_MAX ? c_rpb : RPB_MAX, c_bstride};
  g.ldw = ldw;
  g.beta = beta;
  g.conv_shift = -1; g.conv_dy = 0; g.conv_dx = 0;
  g.sk_ws = sk_workspace; g.sk_minparts = 0;
  g.m_split = 0;
  g.rowwise_ok = side_ok16(N, M, C, ldc, c_rpb, c_bstride, aux, ldaux, gate, gate_ld, bias);
  {
    const long a_last = (long)((M - 1) / a_rpb) * a_bstride + (long)((M - 1) % a_rpb) * lda + K;
    g.span32 = a_last * 2 < (1L << 32) && ((long)N * ldw) * 2 < (1L << 32);
  }
  return dispatch(g, epilogue, (hipStream_t)stream);
}

// Two problems of equal N, K, epilogue and leading dimensions in ONE launch of the persistent kernel: the text- and the
// image-stream Linear of a FLUX double block (diffusers issues them as separate nn.Linear calls,
// fastvideo/utils/sampling_utils.py:68-82).  The tile grid's first M1 rows are problem 1, the rest problem 2 (M1 % 256 == 0),
// so the text stream's few tile rows ride the image stream's rounds instead of occupying a third of the chip for a tile time
// of their own.  The second problem's A / W must lie within 4 GiB of the first's (the same activation buffer / the same block of
// the parameter store).  Whatever the persistent kernel cannot take goes out as two launches of mgx_gemm_bf16_sk: same results.
extern "C" int mgx_gemm_bf16_pair(const uint16_t* A1, const uint16_t* W1, const uint16_t* bias1, void* C1, const uint16_t* gate1,
                                  uint16_t* aux1, int M1, long a1_rpb, long a1_bstride, long c1_rpb, long c1_bstride,
                                  const uint16_t* A2, const uint16_t* W2, const uint16_t* bias2, void* C2, const uint16_t* gate2,
                                  uint16_t* aux2, int M2, long a2_rpb, long a2_bstride, long c2_rpb, long c2_bstride, int N, int K,
                                  long lda, long ldw, long ldc, long ldaux, long gate_ld, int epilogue, float beta,
                                  float* sk_workspace, long sk_workspace_elems, void* stream) {
  MGX_REQUIRE(A1 && W1 && C1 && A2 && W2 && C2, "null operand");
  MGX_REQUIRE((bias1 == nullptr) == (bias2 == nullptr) && (gate1 == nullptr) == (gate2 == nullptr) && (aux1 == nullptr) == (aux2 == nullptr),
              "the two problems must use the same optional operands");
  const long RPB_MAX = 1L << 30;
  // offsets are taken from the LOWER of the two pointers (the image stream's weights precede the text stream's in the store)
  const uint16_t* Ab = A1 < A2 ? A1 : A2;
  const uint16_t* Wb = W1 < W2 ? W1 : W2;
  const long a1o = A1 - Ab, a2o = A2 - Ab, w1o = W1 - Wb, w2o = W2 - Wb;
  bool groupable = M1 > 0 && M2 > 0 && M1 % 256 == 0 && epilogue != EPI_F32_ACC && K % BK == 0 && N % 4 == 0 &&
                   ((uintptr_t)A1 % 16 == 0) && ((uintptr_t)W1 % 16 == 0) && ((uintptr_t)A2 % 16 == 0) && ((uintptr_t)W2 % 16 == 0) &&
                   a2_rpb > 0 && c2_rpb > 0 && a1_rpb > 0 && c1_rpb > 0;
  if (groupable) {
    const long a1_last = a1o + (long)((M1 - 1) / a1_rpb) * a1_bstride + (long)((M1 - 1) % a1_rpb) * lda + K;
    const long a2_last = a2o + (long)((M2 - 1) / a2_rpb) * a2_bstride + (long)((M2 - 1) % a2_rpb) * lda + K;
    groupable = a1_last * 2 < (1L << 32) && a2_last * 2 < (1L << 32) && ((w1o > w2o ? w1o : w2o) + (long)N * ldw) * 2 < (1L << 32);
  }
  if (groupable) {
    MGX_REQUIRE(!sk_workspace || (sk_workspace_elems >= mgx_gemm_sk_workspace_elems() && (uintptr_t)sk_workspace % 16 == 0),
                "stream-K workspace too small (mgx_gemm_sk_workspace_elems) or misaligned");
    MGX_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0 && (!aux1 || ldaux % 4 == 0), "leading dimensions must keep alignment");
    MGX_REQUIRE(epilogue != EPI_BIAS_GATE_RES || gate1, "gate-residual epilogue needs a gate");
    MGX_REQUIRE(epilogue != EPI_BIAS_MULAUX || aux1, "gelu-backward epilogue needs the saved pre-activation");
    GemmArgs g{};
    g.A = Ab; g.W = Wb; g.bias = bias1; g.C = C1; g.gate = gate1; g.aux = aux1; g.ldaux = ldaux; g.gate_ld = gate_ld;
    g.M = M1 + M2; g.N = N; g.K = K;
    g.a = RowMap{lda, a1_rpb < RPB_MAX ? a1_rpb : RPB_MAX, a1_bstride};
    g.c = RowMap{ldc, c1_rpb < RPB_MAX ? c1_rpb : RPB_MAX, c1_bstride};
    g.ldw = ldw; g.beta = beta;
    g.conv_shift = -1; g.conv_dy = 0; g.conv_dx = 0;
    g.sk_ws = sk_workspace; g.sk_minparts = 0;
    g.m_split = M1; g.a1_off = a1o; g.a2_off = a2o; g.w1_off = w1o; g.w2_off = w2o;
    g.a2 = RowMap{lda, a2_rpb < RPB_MAX ? a2_rpb : RPB_MAX, a2_bstride};
    g.c2 = RowMap{ldc, c2_rpb < RPB_MAX ? c2_rpb : RPB_MAX, c2_bstride};
    g.C2 = C2; g.bias2 = bias2; g.gate2 = gate2; g.aux2 = aux2;
    g.rowwise_ok = side_ok16(N, M1, C1, ldc, c1_rpb, c1_bstride, aux1, ldaux, gate1, gate_ld, bias1) &&
                   side_ok16(N, M2, C2, ldc, c2_rpb, c2_bstride, aux2, ldaux, gate2, gate_ld, bias2);
    g.span32 = 1;
    const int rc = dispatch(g, epilogue, (hipStream_t)stream);
    if (rc <= 0) return rc;               // launched (0) or failed (< 0); 1: too small for the persistent kernel
  }
  const int rc = mgx_gemm_bf16_sk(A1, W1, bias1, C1, gate1, aux1, ldaux, M1, N, K, lda, a1_rpb, a1_bstride, ldw, ldc, c1_rpb, c1_bstride,
                                  gate_ld, epilogue, beta, sk_workspace, sk_workspace_elems, stream);
  if (rc) return rc;
  return mgx_gemm_bf16_sk(A2, W2, bias2, C2, gate2, aux2, ldaux, M2, N, K, lda, a2_rpb, a2_bstride, ldw, ldc, c2_rpb, c2_bstride, gate_ld,
                          epilogue, beta, sk_workspace, sk_workspace_elems, stream);
}

// Ct[b][f][t] = bf16(X[b * tok_rpb + t, :] . W[f, :] + bias[f]): a Linear whose output leaves TRANSPOSED, token-contiguous
// (ld_ct elements between feature rows, ct_bstride between token batches) -- the V^T [B, H, 128, Sp] operand of the attention
// kernels straight from the projection, without the transposing pass of mgx_qk_norm_rope_fwd.  Runs the persistent kernel
// with the operand roles swapped (rows = features, columns = tokens; GemmArgs::bias_rows / col_rpb).  Returns 1 -- nothing
// launched -- for what that kernel or its 16-byte epilogue cannot take (few tiles, odd alignments): the caller then keeps the
// plain Linear + transposing pass.
extern "C" int mgx_linear_bf16_t(const uint16_t* X, const uint16_t* W, const uint16_t* bias, uint16_t* Ct, int tokens, int F, int K,
                                 long ldx, long ldw, long ld_ct, long tok_rpb, long ct_bstride, float* sk_workspace,
                                 long sk_workspace_elems, void* stream) {
  MGX_REQUIRE(X && W && Ct && tokens > 0 && F > 0 && K > 0 && tok_rpb > 0, "bad argument");
  MGX_REQUIRE(!sk_workspace || (sk_workspace_elems >= mgx_gemm_sk_workspace_elems() && (uintptr_t)sk_workspace % 16 == 0),
              "stream-K workspace too small (mgx_gemm_sk_workspace_elems) or misaligned");
  const long rpb = tok_rpb < tokens ? tok_rpb : tokens;
  const long nb = (tokens + rpb - 1) / rpb;
  const bool ok = K % BK == 0 && tokens % 8 == 0 && F % 4 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ld_ct % 8 == 0 && ct_bstride % 8 == 0 &&
                  (nb == 1 || rpb % 64 == 0) && ld_ct >= rpb && ((uintptr_t)X % 16 == 0) && ((uintptr_t)W % 16 == 0) &&
                  ((uintptr_t)Ct % 16 == 0) && (!bias || (uintptr_t)bias % 2 == 0);
  if (!ok) return 1;
  GemmArgs g{};
  g.A = W; g.W = X; g.bias = bias; g.C = Ct;
  g.M = F; g.N = tokens; g.K = K;
  g.a = RowMap{ldw, 1L << 30, 0};
  g.c = RowMap{ld_ct, 1L << 30, 0};
  g.ldw = ldx;
  g.conv_shift = -1;
  g.sk_ws = sk_workspace;
  g.rowwise_ok = 1;
  g.bias_rows = 1;
  g.col_rpb = nb > 1 ? rpb : 0; g.col_bstride = ct_bstride;
  g.band = 6;     // (a caller's hint, launch(): bands of 6 feature-tile rows -- 0.522 ms against 0.534 ms for one band of all 12 at
                  //  3072 x 36864 x 3072, profiles/r04_split_projection_band_sweep.log; the weight gradients keep their rule)
  g.span32 = ((long)F * ldw) * 2 < (1L << 32) && ((long)tokens * ldx) * 2 < (1L << 32);
  if (!persistent_ok(g)) return 1;
  return launch<EPI_BIAS>(g, (hipStream_t)stream);
}

// The q | k projection of an attention layer with mgx_qk_norm_rope_fwd_qs applied in the GEMM's epilogue: X [tokens, K] (plain,
// B samples of rows_per_batch tokens each), Wqk [2 * H * 128, K] (the to_q rows, then the to_k rows), bias [2 * H * 128] ->
// Q, K [B, H, S, 128] at sequence positions s0 .. s0 + rows_per_batch - 1, Q times q_scale.  Same bits as the plain Linear
// followed by mgx_qk_norm_rope_fwd_qs.  Returns 1 -- nothing launched -- when the persistent kernel cannot take the problem
// (fewer than 128 tiles of 256 x 256, H odd, rows_per_batch % 128 != 0, alignments): the caller keeps the two-pass form.
extern "C" int mgx_linear_qk_norm_rope(const uint16_t* X, const uint16_t* Wqk, const uint16_t* bias, const float* wq,
                                       const float* wk, const float* cos, const float* sin, const float* cos_sin_pairs,
                                       uint16_t* Q, uint16_t* K, int B, int H, int S, int rows_per_batch, int s0, int Kdim,
                                       long ldx, long ldw, float q_scale, void* stream) {
  MGX_REQUIRE(X && Wqk && wq && wk && cos && sin && Q && K, "null argument");
  MGX_REQUIRE(!cos_sin_pairs || (uintptr_t)cos_sin_pairs % 16 == 0, "pair table must be 16-byte aligned");
  MGX_REQUIRE(B > 0 && H > 0 && rows_per_batch > 0 && s0 >= 0 && s0 + rows_per_batch <= S && Kdim > 0 && q_scale > 0.f, "bad sizes");
  static const int off = getenv("MGX_GEMM_QKNORM") ? atoi(getenv("MGX_GEMM_QKNORM")) == 0 : 0;
  const long tokens = (long)B * rows_per_batch;
  const int dmodel = H * 128;
  const bool ok = !off && dmodel % 256 == 0 && rows_per_batch % 128 == 0 && Kdim % BK == 0 && ldx % 8 == 0 && ldw % 8 == 0 &&
                  tokens < (1L << 31) && ((uintptr_t)X % 16 == 0) && ((uintptr_t)Wqk % 16 == 0) && ((uintptr_t)Q % 16 == 0) &&
                  ((uintptr_t)K % 16 == 0) && ((uintptr_t)cos % 16 == 0) && ((uintptr_t)sin % 16 == 0) &&
                  ((uintptr_t)wq % 16 == 0) && ((uintptr_t)wk % 16 == 0) && (!bias || (uintptr_t)bias % 16 == 0);
  if (!ok) return 1;
  GemmArgs g{};
  g.A = X; g.W = Wqk; g.bias = bias; g.C = Q;
  g.M = (int)tokens; g.N = 2 * dmodel; g.K = Kdim;
  g.a = RowMap{ldx, 1L << 30, 0};
  g.c = RowMap{128, rows_per_batch, 0};
  g.ldw = ldw;
  g.conv_shift = -1;
  g.rowwise_ok = 1;
  g.span32 = (tokens * ldx) * 2 < (1L << 32) && ((long)g.N * ldw) * 2 < (1L << 32);
  g.qn_wq = wq; g.qn_wk = wk; g.qn_cos = cos; g.qn_sin = sin; g.qn_cs2 = cos_sin_pairs; g.qn_Q = Q; g.qn_K = K;
  g.qn_H = H; g.qn_S = S; g.qn_s0 = s0; g.qn_dmodel = dmodel; g.qn_qscale = q_scale;
  if (!persistent_ok(g)) return 1;
  return cos_sin_pairs ? launch<EPI_QKNORM_P>(g, (hipStream_t)stream) : launch<EPI_QKNORM>(g, (hipStream_t)stream);
}

extern "C" int mgx_gemm_bf16(const uint16_t* A, const uint16_t* W, const uint16_t* bias, void* C, const uint16_t* gate,
                             uint16_t* aux, long ldaux, int M, int N, int K, long lda, long a_rpb, long a_bstride, long ldw, long ldc,
                             long c_rpb, long c_bstride, long gate_ld, int epilogue, float beta, void* stream) {
  return mgx_gemm_bf16_sk(A, W, bias, C, gate, aux, ldaux, M, N, K, lda, a_rpb, a_bstride, ldw, ldc, c_rpb, c_bstride, gate_ld,
                          epilogue, beta, nullptr, 0, stream);
}

// 3x3 convolution, stride 1, zero padding 1, as an implicit GEMM on the same kernels: x is a zero-bordered NHWC image
// [(H + 2) x (W + 2) x C] (the border is the padding: no bounds checks in the K-loop), the weight [Cout][3][3][C] (tap-major,
// channels contiguous), out [H W x Cout] row-major with leading dimension ld_out.  residual != 0: out += conv(x) + bias
// (bf16 sum of two bf16 tensors, like `input_tensor + hidden_states` of a ResnetBlock2D), through the gate-residual epilogue
// with a gate of ones.
extern "C" int mgx_conv3x3_nhwc(const uint16_t* x, const uint16_t* Wt, const uint16_t* bias, uint16_t* out, long ld_out,
                                const uint16_t* ones, int H, int Wd, int C, int Cout, int residual, void* stream) {
  MGX_REQUIRE(x && Wt && out && H > 0 && Wd > 0, "bad argument");
  MGX_REQUIRE(C == 64 || C == 128 || C == 256 || C == 512, "channels per tap must be 64, 128, 256 or 512 (pad with zeros)");
  MGX_REQUIRE(Cout > 0 && Cout % 4 == 0 && ld_out >= Cout && ld_out % 4 == 0, "output channels must be a multiple of 4");
  MGX_REQUIRE(!residual || ones, "the residual form needs a vector of Cout ones (bf16)");
  MGX_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)Wt % 16 == 0) && ((uintptr_t)out % 8 == 0), "operands must be 16-byte aligned");
  const long Wp = Wd + 2;
  MGX_REQUIRE((long)(H + 2) * Wp * C * 2 < (1L << 32) && (long)Cout * 9 * C * 2 < (1L << 32), "image too large for one call");
  GemmArgs g{};
  g.A = x; g.W = Wt; g.bias = bias; g.C = out; g.gate = residual ? ones : nullptr; g.aux = nullptr; g.ldaux = 0; g.gate_ld = 0;
  g.M = H * Wd; g.N = Cout; g.K = 9 * C;
  g.a = RowMap{C, Wd, Wp * C};                      // output pixel (y, x) -> padded pixel (y, x): tap (0, 0)
  g.c = RowMap{ld_out, 1L << 30, 0};
  g.ldw = 9L * C;
  g.beta = 0.f;
  g.conv_shift = C == 64 ? 0 : (C == 128 ? 1 : (C == 256 ? 2 : 3));
  g.conv_dy = Wp * C; g.conv_dx = C;
  g.sk_ws = nullptr; g.sk_minparts = 0; g.m_split = 0;
  g.rowwise_ok = (Cout % 8 == 0) && (ld_out % 8 == 0) && ((uintptr_t)out % 16 == 0) && (!bias || (uintptr_t)bias % 16 == 0) &&
                 (!residual || (uintptr_t)ones % 16 == 0);
  g.span32 = 1;
  hipStream_t st = (hipStream_t)stream;
  return residual ? launch<EPI_BIAS_GATE_RES, true>(g, st) : launch<EPI_BIAS, true>(g, st);
}

extern "C" long mgx_transpose_partial_elems(int M, int N) { return (long)cdiv(M, 64) * N; }   // covers both paths

extern "C" int mgx_transpose_bf16(const uint16_t* in, uint16_t* out, float* colsum_partial, float* colsum_out,
                                  float colsum_beta, int M, int N, long ld_in, long in_rpb, long in_bstride, long ld_out,
                                  void* stream) {
  MGX_REQUIRE(in && out && M > 0 && N > 0, "bad argument");
  MGX_REQUIRE(ld_out >= M, "output leading dimension must cover M");
  MGX_REQUIRE((colsum_partial == nullptr) == (colsum_out == nullptr), "column sums need both workspace and output");
  hipStream_t st = (hipStream_t)stream;
  const bool fast = N % 8 == 0 && ld_in % 8 == 0 && in_bstride % 8 == 0 && ld_out % 8 == 0 &&
                    ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0);
  if (fast) {
    dim3 grid(cdiv(N, 128), cdiv(ld_out, 128));
    transpose8_kernel<false><<<grid, 256, 0, st>>>(in, out, colsum_partial, M, N, RowMap{ld_in, in_rpb < (1L << 30) ? in_rpb : (1L << 30), in_bstride}, ld_out);
    if (colsum_out) colsum_finish_kernel<<<cdiv(N, 64), 256, 0, st>>>(colsum_partial, colsum_out, cdiv(M, 128), N, colsum_beta);
  } else {
    dim3 grid(cdiv(N, 64), cdiv(ld_out, 64));
    transpose_kernel<<<grid, 256, 0, st>>>(in, out, colsum_partial, M, N, RowMap{ld_in, in_rpb < (1L << 30) ? in_rpb : (1L << 30), in_bstride}, ld_out);
    if (colsum_out) colsum_finish_kernel<<<cdiv(N, 64), 256, 0, st>>>(colsum_partial, colsum_out, cdiv(M, 64), N, colsum_beta);
  }
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int mgx_transpose_gelu_bf16(const uint16_t* in, uint16_t* out, int M, int N, long ld_in, long ld_out, void* stream) {
  MGX_REQUIRE(in && out && M > 0 && N > 0, "bad argument");
  MGX_REQUIRE(ld_out >= M, "output leading dimension must cover M");
  MGX_REQUIRE(N % 8 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0 && ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0),
              "rows must be 16-byte addressable");
  dim3 grid(cdiv(N, 128), cdiv(ld_out, 128));
  transpose8_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(in, out, nullptr, M, N, RowMap{ld_in, 1L << 30, 0}, ld_out);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int mgx_gelu_bf16(const uint16_t* x, long ldx, uint16_t* y, long ldy, long M, int N, void* stream) {
  MGX_REQUIRE(x && y && M > 0 && N > 0, "bad argument");
  MGX_REQUIRE(N % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0),
              "rows must be 16-byte addressable");
  const long total = M * (N / 8);
  const int grid = (int)((total + 255) / 256 < 256L * 32 ? (total + 255) / 256 : 256L * 32);
  gelu_rows_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, ldx, y, ldy, M, N / 8);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}
