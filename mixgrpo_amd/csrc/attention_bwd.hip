// Joint attention backward (non-causal, head_dim 128) for the FLUX MMDiT replay pass.
//
// Replaces the autograd of F.scaled_dot_product_attention (flash-attention backward in the reference's stack).
// P is recomputed from Q, K and the forward's LSE.  Two kernels, no atomics, bitwise reproducible:
//   * attn_bwd_dkv : one wave owns 32 keys (K, V fragments in registers), the workgroup (8 waves = 256 keys)
//                    sweeps 32-query tiles;  S = Q K^T and dP = dO V^T with the KEY on the MFMA lane, so their
//                    accumulators are directly the B operands of dV^T += dO^T P and dK^T += Q^T dS.
//   * attn_bwd_dq  : one wave owns 32 queries (Q, dO fragments in registers), the workgroup (256 queries) sweeps
//                    64-key tiles;  S^T = K Q^T, dP^T = V dO^T with the QUERY on the lane, dQ^T += K^T dS^T.
// Operands that are consumed "k-strided" by the second product of each pair are provided pre-transposed in
// global memory ([B,H,128,Sp]: Qt, Kt by qk_norm_rope, dOt by attn_bwd_prep), so every LDS fragment read is a
// plain conflict-free ds_read (XOR-swizzled images), cf. the forward kernel.
#include "../../include/mixgrpo_hip.h"
#include "common.h"

#include <cstdlib>
#include <type_traits>

namespace {

constexpr int HD = 128;

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));   // one v_cvt_pk_bf16_f32
}
// row-major [rows][128] bf16 image, 16-byte chunk swizzle
__device__ __forceinline__ int rm_off(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }
// Transposed images [128 d][64 or 32 columns] in 16-byte SLOTS: slot (b, h) of a row holds, for the 16-column block b,
// the columns 4h+{0..3} and 4h+8+{0..3} -- the 8 columns (in the S / dS accumulator's order) that lane half h feeds to
// one MFMA, so every transposed A operand is ONE conflict-free ds_read_b128.  (Two ds_read_b64 per operand are fused by
// hipcc into ds_read2st64_b64: half the LDS rate, modulo-32 banking, 2-way conflicts; see attention.hip.)
// 64 columns: 128-byte rows, 8 slots, slot ^= (d >> 1) & 7.   32 columns: 64-byte rows, 4 slots, slot ^= ((d >> 2) ^ (d >> 1)) & 3
// (both checked exhaustively against the ds_read_b128 lane groups and the ds_write_b64 groups).
__device__ __forceinline__ int t64_off(int d, int slot) { return d * 128 + ((slot ^ ((d >> 1) & 7)) << 4); }
__device__ __forceinline__ int t32_off(int d, int slot) { return d * 64 + ((slot ^ (((d >> 2) ^ (d >> 1)) & 3)) << 4); }

// Staging loads go through buffer descriptors (wave-uniform base in SGPRs + ONE loop-invariant 32-bit lane offset + a scalar
// tile offset): the 64-bit per-lane pointers of plain loads cost ~2 VGPRs each in kernels that sit at the 256-register
// limit (their spill reloads inside the tile loop also drain the load queue), and rows beyond the tensor read as zeros
// (out-of-range records) instead of needing a clamp per tile.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__device__ __forceinline__ auto buf_rsrc(const void* p, long bytes) {
  const unsigned n = bytes > 0xfffffff0L ? 0xfffffff0u : (unsigned)bytes;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)n, 0x00020000);
}
template <typename R>
__device__ __forceinline__ uint4 buf_ld16(R rs, unsigned voff, unsigned soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}
template <typename R>
__device__ __forceinline__ float buf_ld_f32(R rs, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)soff, 0));
}

__device__ __forceinline__ void xcd_remap(int& bid, int nwg) {
  const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
}

// ------------------------------------------------------------------------------------------------ prep
// delta[b,h,s] = sum_d dO[b,s,h*128+d] * O[b,s,h*128+d];  dOt[b,h,d,s] = dO[b,s,h*128+d]
__global__ void __launch_bounds__(256) attn_bwd_prep_kernel(const bf16_raw* __restrict__ O, const bf16_raw* __restrict__ dO,
                                                            long ldo, long o_bstride, float* __restrict__ delta,
                                                            bf16_raw* __restrict__ dOt, int H, int S, int Sp) {
  __shared__ bf16_raw tile[64][130];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int hh = blockIdx.y, b = blockIdx.z, t0 = blockIdx.x * 64;
  for (int i = 0; i < 16; ++i) {
    const int tl = w * 16 + i, s = t0 + tl;
    uint32_t ud = 0;
    float acc = 0.f;
    if (s < S) {
      const long off = (long)b * o_bstride + (long)s * ldo + hh * HD + 2 * lane;
      ud = *reinterpret_cast<const uint32_t*>(dO + off);
      const uint32_t uo = *reinterpret_cast<const uint32_t*>(O + off);
      acc = bf2f(ud & 0xffff) * bf2f(uo & 0xffff) + bf2f(ud >> 16) * bf2f(uo >> 16);
    }
    *reinterpret_cast<uint32_t*>(&tile[tl][2 * lane]) = ud;
    acc = wave_sum(acc);
    if (lane == 0 && s < S) delta[((long)b * H + hh) * S + s] = acc;
  }
  __syncthreads();
  for (int id = threadIdx.x; id < 128 * 8; id += 256) {
    const int d = id >> 3, c = id & 7;
    bf16_raw* dst = dOt + (((long)b * H + hh) * HD + d) * Sp + t0 + c * 8;
    if (t0 + c * 8 < Sp) {
      uint4 u;
      u.x = (uint32_t)tile[c * 8 + 0][d] | ((uint32_t)tile[c * 8 + 1][d] << 16);
      u.y = (uint32_t)tile[c * 8 + 2][d] | ((uint32_t)tile[c * 8 + 3][d] << 16);
      u.z = (uint32_t)tile[c * 8 + 4][d] | ((uint32_t)tile[c * 8 + 5][d] << 16);
      u.w = (uint32_t)tile[c * 8 + 6][d] | ((uint32_t)tile[c * 8 + 7][d] << 16);
      *reinterpret_cast<uint4*>(dst) = u;   // rows beyond S hold zeros (ud = 0)
    }
  }
}

struct BwdArgs {
  const bf16_raw* Q;    // [B,H,S,128]
  const bf16_raw* K;    // [B,H,S,128]
  const bf16_raw* V;    // [B,H,S,128]
  const bf16_raw* Qt;   // [B,H,128,Sp]
  const bf16_raw* Kt;   // [B,H,128,Sp]
  const bf16_raw* dO;   // [B,S,ldo] at column h*128
  const bf16_raw* dOt;  // [B,H,128,Sp]
  const float* lse;     // [B,H,S]
  const float* delta;   // [B,H,S]
  bf16_raw* dQ;         // [B,H,S,128]
  bf16_raw* dK;
  bf16_raw* dV;
  int B, H, S, Sp;
  long ldo, o_bstride;
  float scale, scale_log2e, neg_inv_scale;
};

// ------------------------------------------------------------------------------------------------ dK, dV
// LDS per q-tile (32 queries): Q rm [32][128] 8K | dO rm 8K | Qt t32 [128][32] 8K | dOt t32 8K | lse 128 B | delta 128 B
constexpr int DKV_STAGE = 4 * 8192 + 256;
__global__ void __launch_bounds__(512, 2) attn_bwd_dkv_kernel(BwdArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nkb = (g.S + 255) / 256;
  int bid = blockIdx.x;
  xcd_remap(bid, nkb * g.H * g.B);
  const int kt = bid % nkb, bh = bid / nkb;
  const int b = bh / g.H, hh = bh - b * g.H;
  const long bhS = (long)bh * g.S;
  const bf16_raw* Qp = g.Q + bhS * HD;
  const bf16_raw* Qtp = g.Qt + (long)bh * HD * g.Sp;
  const bf16_raw* dOtp = g.dOt + (long)bh * HD * g.Sp;
  const bf16_raw* dOp = g.dO + (long)b * g.o_bstride + hh * HD;

  // own keys: K and V fragments as B operands: B[k = d][col = key]  -> K[key0 + r][16ks + 8h + j]
  const int key0 = kt * 256 + wid * 32;
  int krow = key0 + r;
  const bool key_valid = krow < g.S;
  if (krow >= g.S) krow = g.S - 1;
  // K fragments stay in registers; the V fragments live in a wave-private LDS area ([ks][lane] 16-byte pieces, read
  // back lane-linear): with both in registers the kernel needs ~290 VGPRs and hipcc spills 32 of them to scratch and
  // reloads them INSIDE the q-tile loop, where each reload's vmcnt wait also drains the staging loads issued early.
  s16x8 kf[8];
  char* vfrag = smem + 2 * DKV_STAGE + wid * 8192 + lane * 16;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    kf[ks] = *reinterpret_cast<const s16x8*>(g.K + (bhS + krow) * HD + ks * 16 + h * 8);
    *reinterpret_cast<s16x8*>(vfrag + ks * 1024) =
        *reinterpret_cast<const s16x8*>(g.V + (bhS + krow) * HD + ks * 16 + h * 8);
  }
  f32x16 dv[4], dk[4];   // dV^T, dK^T tiles: rows d = 32*dt + ..., column = key r
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) dv[dt][i] = dk[dt][i] = 0.f;

  // staging roles: 512 threads; Q rm: 512 chunks (row = id>>4, chunk = id&15); dO rm same; Qt/dOt t32: 512 16-B
  // chunks each (d = id>>2, c16 = id&3 -> two 8-byte chunks)
  const int s_row = tid >> 4, s_chunk = tid & 15;
  const int s_d = tid >> 2, s_c16 = tid & 3;
  uint4 rq, rdo, rqt, rdot;
  float rl = 0.f;
  const int nqt = (g.S + 31) / 32;
  const auto rs_q = buf_rsrc(Qp, (long)g.S * HD * 2);                               // rows >= S read as zeros
  const auto rs_do = buf_rsrc(dOp, ((long)(g.S - 1) * g.ldo + HD) * 2);
  const auto rs_qt = buf_rsrc(Qtp, (long)HD * g.Sp * 2);
  const auto rs_dot = buf_rsrc(dOtp, (long)HD * g.Sp * 2);
  const auto rs_lse = buf_rsrc(g.lse + bhS, (long)g.S * 4);
  const auto rs_dlt = buf_rsrc(g.delta + bhS, (long)g.S * 4);
  const unsigned vo_q = (unsigned)(s_row * HD + s_chunk * 8) * 2, vo_do = (unsigned)(s_row * g.ldo + s_chunk * 8) * 2;
  const unsigned vo_t = (unsigned)(s_d * g.Sp + s_c16 * 8) * 2, vo_l = (unsigned)(tid & 31) * 4;
  float rl2 = 0.f;
  const bool wave0 = __builtin_amdgcn_readfirstlane(wid) == 0;
  const float inv_scale = 1.0f / g.scale;
#define DKV_LOAD(t)                                                                                   \
  do {                                                                                                \
    const unsigned q_base = (unsigned)(t) * 32;                                                       \
    rq = buf_ld16(rs_q, vo_q, q_base * (HD * 2));                                                     \
    rdo = buf_ld16(rs_do, vo_do, q_base * (unsigned)(g.ldo * 2));                                     \
    rqt = buf_ld16(rs_qt, vo_t, q_base * 2);                                                          \
    rdot = buf_ld16(rs_dot, vo_t, q_base * 2);                                                        \
    /* lse and delta of the tile's 32 queries: loaded unconditionally by every thread (both, 4 bytes each) and only   \
       turned into the staged value at DKV_STORE time.  (A load inside `if (tid < 64) { ... }` with arithmetic on its  \
       result made hipcc wait vmcnt(0) right there: the four staging loads just issued were drained at the top of     \
       EVERY q-tile, ~1 us of a 2.6 us tile.) */                                                      \
    if (wave0) {   /* wave-uniform (scalar branch): the other seven waves issue four loads, not six */        \
      rl = buf_ld_f32(rs_lse, vo_l, q_base * 4);                                                      \
      rl2 = buf_ld_f32(rs_dlt, vo_l, q_base * 4);                                                     \
    }                                                                                                 \
  } while (0)
#define DKV_STORE(buf)                                                                                \
  do {                                                                                                \
    char* base = smem + (buf) * DKV_STAGE;                                                            \
    *reinterpret_cast<uint4*>(base + rm_off(s_row, s_chunk)) = rq;                                    \
    *reinterpret_cast<uint4*>(base + 8192 + rm_off(s_row, s_chunk)) = rdo;                            \
    /* 16-byte chunk s_c16 = queries 8c..8c+7: 4-query groups 2c (h = 0) and 2c+1 (h = 1) of block c >> 1 */ \
    *reinterpret_cast<uint2*>(base + 16384 + t32_off(s_d, (s_c16 >> 1) * 2) + (s_c16 & 1) * 8) = make_uint2(rqt.x, rqt.y);     \
    *reinterpret_cast<uint2*>(base + 16384 + t32_off(s_d, (s_c16 >> 1) * 2 + 1) + (s_c16 & 1) * 8) = make_uint2(rqt.z, rqt.w); \
    *reinterpret_cast<uint2*>(base + 24576 + t32_off(s_d, (s_c16 >> 1) * 2) + (s_c16 & 1) * 8) = make_uint2(rdot.x, rdot.y);   \
    *reinterpret_cast<uint2*>(base + 24576 + t32_off(s_d, (s_c16 >> 1) * 2 + 1) + (s_c16 & 1) * 8) = make_uint2(rdot.z, rdot.w); \
    /* staged as the INITIAL ACCUMULATORS of the S and dP chains: -lse / scale (rows beyond S: -inf -> P = 0) and      \
       -delta; S' = Q K^T - lse / scale then gives P = exp2(scale_log2e * S') with no subtraction and no LDS read in the \
       VALU phase (in-kernel stamps: that phase was 30 % of a q-tile, most of it waiting for these broadcast reads) */  \
    if (wave0)                                                                                        \
      reinterpret_cast<float*>(base + 32768)[tid] =                                                   \
          (tid < 32) ? ((nt_ * 32 + tid < g.S) ? -rl * inv_scale : -INFINITY) : -rl2;                 \
  } while (0)

  // (Measured and dropped: `s_setprio 1` for the second-dispatched half of the waves, which loses the VALU arbitration to
  // the older half in every phase -- stamps: 1476 vs 1048 cycles in the exp / dS phase -- either statically or alternating
  // inside a tile: the favoured half simply takes over the other's timing, 7.6 ms either way.)
  DKV_LOAD(0);
  { const int nt_ = 0; DKV_STORE(0); }
  __syncthreads();
  // see attention.hip: make the pre-loop fragment loads provably complete so the loop's MFMAs are not fenced behind
  // the staging loads
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) asm volatile("" :: "v"(kf[ks]));
  int cur = 0;
#define STAMP(k_) do {} while (0)
  auto tile = [&](int t, auto mask_tag) __attribute__((always_inline)) {
    constexpr bool MASK = decltype(mask_tag)::value;
    STAMP(0);
    const char* base = smem + cur * DKV_STAGE;
    const float* nls = reinterpret_cast<const float*>(base + 32768);     // -lse / scale per query row of the tile
    const float* ndl = nls + 32;                                         // -delta
    f32x16 s, dp;
#pragma unroll
    for (int j = 0; j < 4; ++j) {        // accumulator register 4j + e holds query row 8j + 4h + e
      const float4 a = *reinterpret_cast<const float4*>(nls + 8 * j + 4 * h);
      const float4 b2 = *reinterpret_cast<const float4*>(ndl + 8 * j + 4 * h);
      s[4 * j] = a.x; s[4 * j + 1] = a.y; s[4 * j + 2] = a.z; s[4 * j + 3] = a.w;
      dp[4 * j] = b2.x; dp[4 * j + 1] = b2.y; dp[4 * j + 2] = b2.z; dp[4 * j + 3] = b2.w;
    }
    // Fragment reads run ONE k-step ahead of their MFMAs, in two register sets, the order pinned: written as
    // "read; read; read; mfma; mfma" per k-step hipcc waits for the reads it has just issued (lgkmcnt(0) in front of
    // every MFMA pair), so each k-step exposed the full LDS latency and the matrix pipe idled between pairs.
    {
      s16x8 qa[2], da[2], vk;                     // (a second set of V fragments spills a K fragment into the loop)
      qa[0] = *reinterpret_cast<const s16x8*>(base + rm_off(r, h));
      da[0] = *reinterpret_cast<const s16x8*>(base + 8192 + rm_off(r, h));
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        vk = *reinterpret_cast<const s16x8*>(vfrag + ks * 1024);          // needed by the SECOND MFMA of this k-step
        if (ks + 1 < 8) {
          qa[(ks + 1) & 1] = *reinterpret_cast<const s16x8*>(base + rm_off(r, (ks + 1) * 2 + h));
          da[(ks + 1) & 1] = *reinterpret_cast<const s16x8*>(base + 8192 + rm_off(r, (ks + 1) * 2 + h));
        }
        __builtin_amdgcn_sched_barrier(0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[ks & 1], kf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da[ks & 1], vk, dp, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // next q-tile's staging loads: issued HERE, not at the top of the tile -- their 18 registers would be live through the
    // S / dP phase, the register peak of the kernel (one K fragment was spilled into the loop); Q / dO tiles are L2 hits
    // (every key block of the (batch, head) reads them) and have the exp and dV / dK phases to land
    if (t + 1 < nqt) DKV_LOAD(t + 1);
    STAMP(1);
    // P[q][key] and dS[q][key]; q = (i&3) + 8*(i>>2) + 4h (rows), key = this lane's column
    uint32_t pb[8], dsb[8];
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      float p0 = __builtin_amdgcn_exp2f(s[i] * g.scale_log2e);                  // rows beyond S: S' = -inf -> P = 0
      float p1 = __builtin_amdgcn_exp2f(s[i + 1] * g.scale_log2e);
      if (MASK && !key_valid) p0 = p1 = 0.f;
      const float d0 = p0 * dp[i] * g.scale;
      const float d1 = p1 * dp[i + 1] * g.scale;
      pb[i >> 1] = pack_bf16(p0, p1);
      dsb[i >> 1] = pack_bf16(d0, d1);
    }
    STAMP(2);
    // dV^T += dO^T P, dK^T += Q^T dS: 16 MFMAs, each with ONE operand from LDS (dOt / Qt images).  The reads run two
    // MFMAs ahead through a ring of three fragment registers (same reason as above: "read; mfma" pairs serialise into
    // read -> lgkmcnt(0) -> MFMA, one exposed LDS round trip per MFMA).
    {
      s16x8 pf[2], df[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        pf[s2] = __builtin_bit_cast(s16x8, make_uint4(pb[4 * s2], pb[4 * s2 + 1], pb[4 * s2 + 2], pb[4 * s2 + 3]));
        df[s2] = __builtin_bit_cast(s16x8, make_uint4(dsb[4 * s2], dsb[4 * s2 + 1], dsb[4 * s2 + 2], dsb[4 * s2 + 3]));
      }
      // MFMA i (0..15): s2 = i >> 3, dt = (i >> 1) & 3, which = i & 1 (0: dV from the dOt image, 1: dK from the Qt image)
      auto frag = [&](int i) __attribute__((always_inline)) {
        const int s2 = i >> 3, dt = (i >> 1) & 3, which = i & 1;
        return *reinterpret_cast<const s16x8*>(base + (which ? 16384 : 24576) + t32_off(dt * 32 + r, s2 * 2 + h));
      };
      s16x8 fr[3];                                // (a ring of four fragments spills one K fragment into the loop)
      fr[0] = frag(0); fr[1] = frag(1);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i + 2 < 16) fr[(i + 2) % 3] = frag(i + 2);
        __builtin_amdgcn_sched_barrier(0);
        const int s2 = i >> 3, dt = (i >> 1) & 3;
        if (i & 1) dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % 3], df[s2], dk[dt], 0, 0, 0);
        else dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i % 3], pf[s2], dv[dt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    STAMP(3);
    if (t + 1 < nqt) { const int nt_ = t + 1; DKV_STORE(cur ^ 1); }
    STAMP(4);
    __syncthreads();
    STAMP(5);
    cur ^= 1;
  };
  if (key0 + 32 > g.S) {          // wave-uniform: only the ragged last key block carries the per-key mask
    for (int t = 0; t < nqt; ++t) tile(t, std::true_type{});
  } else {
    for (int t = 0; t < nqt; ++t) tile(t, std::false_type{});
  }
#undef STAMP
  const int key = key0 + r;
  if (key < g.S) {
    bf16_raw* dkp = g.dK + (bhS + key) * HD;
    bf16_raw* dvp = g.dV + (bhS + key) * HD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        const int d = dt * 32 + 8 * i4 + 4 * h;
        *reinterpret_cast<uint2*>(dkp + d) = make_uint2(pack_bf16(dk[dt][4 * i4], dk[dt][4 * i4 + 1]),
                                                        pack_bf16(dk[dt][4 * i4 + 2], dk[dt][4 * i4 + 3]));
        *reinterpret_cast<uint2*>(dvp + d) = make_uint2(pack_bf16(dv[dt][4 * i4], dv[dt][4 * i4 + 1]),
                                                        pack_bf16(dv[dt][4 * i4 + 2], dv[dt][4 * i4 + 3]));
      }
  }
}

// ------------------------------------------------------------------------------------------------ dQ
// LDS per key tile (64 keys): K rm [64][128] 16K | V rm 16K | Kt t64 [128][64] 16K
constexpr int DQ_STAGE = 3 * 16384;
__global__ void __launch_bounds__(512, 2) attn_bwd_dq_kernel(BwdArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int nq = (g.S + 255) / 256;
  int bid = blockIdx.x;
  xcd_remap(bid, nq * g.H * g.B);
  const int qt = bid % nq, bh = bid / nq;
  const int b = bh / g.H, hh = bh - b * g.H;
  const long bhS = (long)bh * g.S;
  const bf16_raw* Kp = g.K + bhS * HD;
  const bf16_raw* Vp = g.V + bhS * HD;
  const bf16_raw* Ktp = g.Kt + (long)bh * HD * g.Sp;

  const int q0 = qt * 256 + wid * 32;
  int qrow = q0 + r;
  const bool q_valid = qrow < g.S;
  if (qrow >= g.S) qrow = g.S - 1;
  s16x8 qf[8], dof[8];
  const bf16_raw* dOp = g.dO + (long)b * g.o_bstride + (long)qrow * g.ldo + hh * HD;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    qf[ks] = *reinterpret_cast<const s16x8*>(g.Q + (bhS + qrow) * HD + ks * 16 + h * 8);
    dof[ks] = *reinterpret_cast<const s16x8*>(dOp + ks * 16 + h * 8);
  }
  // (the dK / dV kernel's "row constants as initial accumulators" does not pay here: the constants are per-lane scalars
  // already, and 32 accumulator registers that are live before their chains cost this kernel its staging registers)
  const float lse2 = g.lse[bhS + qrow] * 1.4426950408889634f;
  const float dlt = g.delta[bhS + qrow];
  f32x16 dq[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) dq[dt][i] = 0.f;

  const int kc_key0 = tid >> 4, kc_chunk = tid & 15;
  const int vc_d0 = tid >> 3, vc_chunk = tid & 7;
  uint4 sk0, sk1, sv0, sv1, st0, st1;
  const int ntiles = (g.S + 63) / 64;
  const auto rs_k = buf_rsrc(Kp, (long)g.S * HD * 2);                               // rows >= S read as zeros (keys masked below)
  const auto rs_v = buf_rsrc(Vp, (long)g.S * HD * 2);
  const auto rs_kt = buf_rsrc(Ktp, (long)HD * g.Sp * 2);
  const unsigned vo_k = (unsigned)(kc_key0 * HD + kc_chunk * 8) * 2, vo_kt = (unsigned)(vc_d0 * g.Sp + vc_chunk * 8) * 2;
#define DQ_LOAD(t)                                                                                  \
  do {                                                                                              \
    const unsigned key_base = (unsigned)(t) * 64;                                                   \
    sk0 = buf_ld16(rs_k, vo_k, key_base * (HD * 2));                                                \
    sk1 = buf_ld16(rs_k, vo_k, (key_base + 32) * (HD * 2));                                         \
    sv0 = buf_ld16(rs_v, vo_k, key_base * (HD * 2));                                                \
    sv1 = buf_ld16(rs_v, vo_k, (key_base + 32) * (HD * 2));                                         \
    st0 = buf_ld16(rs_kt, vo_kt, key_base * 2);                                                     \
    st1 = buf_ld16(rs_kt, vo_kt + (unsigned)(64 * g.Sp * 2), key_base * 2);                         \
  } while (0)
#define DQ_STORE(buf)                                                                               \
  do {                                                                                              \
    char* base = smem + (buf) * DQ_STAGE;                                                           \
    *reinterpret_cast<uint4*>(base + rm_off(kc_key0, kc_chunk)) = sk0;                              \
    *reinterpret_cast<uint4*>(base + rm_off(kc_key0 + 32, kc_chunk)) = sk1;                         \
    *reinterpret_cast<uint4*>(base + 16384 + rm_off(kc_key0, kc_chunk)) = sv0;                      \
    *reinterpret_cast<uint4*>(base + 16384 + rm_off(kc_key0 + 32, kc_chunk)) = sv1;                 \
    /* 16-byte chunk vc_chunk = keys 8c..8c+7: 4-key groups 2c (h = 0) and 2c+1 (h = 1) of block c >> 1 */ \
    *reinterpret_cast<uint2*>(base + 32768 + t64_off(vc_d0, (vc_chunk >> 1) * 2) + (vc_chunk & 1) * 8) = make_uint2(st0.x, st0.y); \
    *reinterpret_cast<uint2*>(base + 32768 + t64_off(vc_d0, (vc_chunk >> 1) * 2 + 1) + (vc_chunk & 1) * 8) = make_uint2(st0.z, st0.w); \
    *reinterpret_cast<uint2*>(base + 32768 + t64_off(vc_d0 + 64, (vc_chunk >> 1) * 2) + (vc_chunk & 1) * 8) = make_uint2(st1.x, st1.y); \
    *reinterpret_cast<uint2*>(base + 32768 + t64_off(vc_d0 + 64, (vc_chunk >> 1) * 2 + 1) + (vc_chunk & 1) * 8) = make_uint2(st1.z, st1.w); \
  } while (0)

  DQ_LOAD(0);
  DQ_STORE(0);
  __syncthreads();
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) asm volatile("" :: "v"(qf[ks]), "v"(dof[ks]));
  asm volatile("" :: "v"(lse2), "v"(dlt));
  int cur = 0;
  auto tile = [&](int t, auto mask_tag) __attribute__((always_inline)) {
    constexpr bool MASK = decltype(mask_tag)::value;
    if (t + 1 < ntiles) DQ_LOAD(t + 1);
    const char* base = smem + cur * DQ_STAGE;
    const int key_base = t * 64;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 s, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) s[i] = dp[i] = 0.f;
      // fragment reads two k-steps ahead of their MFMAs, three register sets, order pinned (see attn_bwd_dkv_kernel)
      {
        s16x8 ka[3], va[3];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          ka[ks] = *reinterpret_cast<const s16x8*>(base + rm_off(kb * 32 + r, ks * 2 + h));
          va[ks] = *reinterpret_cast<const s16x8*>(base + 16384 + rm_off(kb * 32 + r, ks * 2 + h));
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          if (ks + 2 < 8) {
            ka[(ks + 2) % 3] = *reinterpret_cast<const s16x8*>(base + rm_off(kb * 32 + r, (ks + 2) * 2 + h));
            va[(ks + 2) % 3] = *reinterpret_cast<const s16x8*>(base + 16384 + rm_off(kb * 32 + r, (ks + 2) * 2 + h));
          }
          __builtin_amdgcn_sched_barrier(0);
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[ks % 3], qf[ks], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[ks % 3], dof[ks], dp, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      uint32_t dsb[8];
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        const int k0 = key_base + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        float p0 = __builtin_amdgcn_exp2f(s[i] * g.scale_log2e - lse2);
        float p1 = __builtin_amdgcn_exp2f(s[i + 1] * g.scale_log2e - lse2);
        if (MASK) {
          if (k0 >= g.S) p0 = 0.f;
          if (k0 + 1 >= g.S) p1 = 0.f;
        }
        dsb[i >> 1] = pack_bf16(p0 * (dp[i] - dlt) * g.scale, p1 * (dp[i + 1] - dlt) * g.scale);
      }
      // dQ^T += K^T dS^T: 8 MFMAs with their Kt fragments three ahead in a ring of four
      {
        s16x8 df[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
          df[s2] = __builtin_bit_cast(s16x8, make_uint4(dsb[4 * s2], dsb[4 * s2 + 1], dsb[4 * s2 + 2], dsb[4 * s2 + 3]));
        auto frag = [&](int i) __attribute__((always_inline)) {       // i = 4 s2 + dt; keys 32 kb + 16 s2 + 4h + {0..3} and + 8
          return *reinterpret_cast<const s16x8*>(base + 32768 + t64_off((i & 3) * 32 + r, (kb * 2 + (i >> 2)) * 2 + h));
        };
        s16x8 fr[4];
        fr[0] = frag(0); fr[1] = frag(1); fr[2] = frag(2);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (i + 3 < 8) fr[(i + 3) & 3] = frag(i + 3);
          __builtin_amdgcn_sched_barrier(0);
          dq[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i & 3], df[i >> 2], dq[i & 3], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (t + 1 < ntiles) DQ_STORE(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  };
  const int nfull = g.S / 64;
  for (int t = 0; t < nfull; ++t) tile(t, std::false_type{});
  if (nfull < ntiles) tile(nfull, std::true_type{});
  if (q_valid) {
    bf16_raw* dqp = g.dQ + (bhS + q0 + r) * HD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        const int d = dt * 32 + 8 * i4 + 4 * h;
        *reinterpret_cast<uint2*>(dqp + d) = make_uint2(pack_bf16(dq[dt][4 * i4], dq[dt][4 * i4 + 1]),
                                                        pack_bf16(dq[dt][4 * i4 + 2], dq[dt][4 * i4 + 3]));
      }
  }
}

// ------------------------------------------------------------------------------------------------ dQ, 64-query waves
// attn_bwd_dq64_kernel: 4 waves x 64 queries (two 32-query chains sharing every K / V / K^T fragment), one wave per SIMD, a
// generated hand-placed instruction stream: csrc/gen/attn_bwd_dq64.py (design, register map, schedule), checked on the CPU by
// tests/test_attn_bwd64_emulated.py.  S % 256 == 0; other shapes take attn_bwd_dq_kernel above.
#include "attn_bwd_dq64_body.inc"

__global__ void __launch_bounds__(256, 1) attn_bwd_dq64_kernel(BwdArgs g) {
  const int nq = g.S >> 8;
  int bid = blockIdx.x;
  xcd_remap(bid, nq * g.H * g.B);
  const int qt = bid % nq, bh = bid / nq;
  const int b = bh / g.H, hh = bh - b * g.H;
  const long bhS = (long)bh * g.S;
  const unsigned long long qp = (unsigned long long)(g.Q + (bhS + qt * 256) * HD);
  const unsigned long long kp = (unsigned long long)(g.K + bhS * HD);
  const unsigned long long vp = (unsigned long long)(g.V + bhS * HD);
  const unsigned long long ktp = (unsigned long long)(g.Kt + (long)bh * HD * g.Sp);
  const unsigned long long dop = (unsigned long long)(g.dO + (long)b * g.o_bstride + (long)(qt * 256) * g.ldo + hh * HD);
  const unsigned long long lsep = (unsigned long long)(g.lse + bhS + qt * 256);
  const unsigned long long dlp = (unsigned long long)(g.delta + bhS + qt * 256);
  const unsigned long long dqp = (unsigned long long)(g.dQ + (bhS + qt * 256) * HD);
  const int ntiles = g.S >> 6;
#define LOHI(x) "s"((unsigned)(x)), "s"((unsigned)((x) >> 32))
  asm volatile(ATTN_BWD_DQ64_BODY
               :
               : [tid] "v"(threadIdx.x), [q_lo] "s"((unsigned)qp), [q_hi] "s"((unsigned)(qp >> 32)), [k_lo] "s"((unsigned)kp),
                 [k_hi] "s"((unsigned)(kp >> 32)), [v_lo] "s"((unsigned)vp), [v_hi] "s"((unsigned)(vp >> 32)),
                 [kt_lo] "s"((unsigned)ktp), [kt_hi] "s"((unsigned)(ktp >> 32)), [do_lo] "s"((unsigned)dop),
                 [do_hi] "s"((unsigned)(dop >> 32)), [lse_lo] "s"((unsigned)lsep), [lse_hi] "s"((unsigned)(lsep >> 32)),
                 [dl_lo] "s"((unsigned)dlp), [dl_hi] "s"((unsigned)(dlp >> 32)), [dq_lo] "s"((unsigned)dqp),
                 [dq_hi] "s"((unsigned)(dqp >> 32)), [sp2] "s"(g.Sp * 2), [ldo2] "s"((int)(g.ldo * 2)), [cs] "s"(g.scale_log2e),
                 [scale] "s"(g.scale), [nloop] "s"((ntiles - 2) >> 1), [seq] "s"(g.S)
               : ATTN_BWD_DQ64_CLOBBERS);
#undef LOHI
}

// ------------------------------------------------------------------------------------------------ dK, dV, 64-key waves
// attn_bwd_dkv64_kernel: 4 waves x 64 keys (two 32-key chains sharing every Q / dO / Q^T / dO^T fragment), one wave per SIMD,
// dV^T and dK^T of both chains in all 256 accumulator registers; generated stream: csrc/gen/attn_bwd_dkv64.py.
#include "attn_bwd_dkv64_body.inc"

__global__ void __launch_bounds__(256, 1) attn_bwd_dkv64_kernel(BwdArgs g) {
  const int nkb = g.S >> 8;
  int bid = blockIdx.x;
  xcd_remap(bid, nkb * g.H * g.B);
  const int kt = bid % nkb, bh = bid / nkb;
  const int b = bh / g.H, hh = bh - b * g.H;
  const long bhS = (long)bh * g.S;
  const unsigned long long qp = (unsigned long long)(g.Q + bhS * HD);
  const unsigned long long dop = (unsigned long long)(g.dO + (long)b * g.o_bstride + hh * HD);
  const unsigned long long qtp = (unsigned long long)(g.Qt + (long)bh * HD * g.Sp);
  const unsigned long long dotp = (unsigned long long)(g.dOt + (long)bh * HD * g.Sp);
  const unsigned long long kp = (unsigned long long)(g.K + (bhS + kt * 256) * HD);
  const unsigned long long vp = (unsigned long long)(g.V + (bhS + kt * 256) * HD);
  const unsigned long long lsep = (unsigned long long)(g.lse + bhS);
  const unsigned long long dlp = (unsigned long long)(g.delta + bhS);
  const unsigned long long dkp = (unsigned long long)(g.dK + (bhS + kt * 256) * HD);
  const unsigned long long dvp = (unsigned long long)(g.dV + (bhS + kt * 256) * HD);
  const int nq = g.S >> 5;
#define LO(x) "s"((unsigned)(x))
#define HI(x) "s"((unsigned)((x) >> 32))
  asm volatile(ATTN_BWD_DKV64_BODY
               :
               : [tid] "v"(threadIdx.x), [q_lo] LO(qp), [q_hi] HI(qp), [do_lo] LO(dop), [do_hi] HI(dop), [qt_lo] LO(qtp),
                 [qt_hi] HI(qtp), [dot_lo] LO(dotp), [dot_hi] HI(dotp), [k_lo] LO(kp), [k_hi] HI(kp), [v_lo] LO(vp), [v_hi] HI(vp),
                 [lse_lo] LO(lsep), [lse_hi] HI(lsep), [dl_lo] LO(dlp), [dl_hi] HI(dlp), [dk_lo] LO(dkp), [dk_hi] HI(dkp),
                 [dv_lo] LO(dvp), [dv_hi] HI(dvp), [sp2] "s"(g.Sp * 2), [ldo2] "s"((int)(g.ldo * 2)), [cs] "s"(g.scale_log2e),
                 [scale] "s"(g.scale), [nis] "s"(g.neg_inv_scale), [nloop] "s"((nq - 2) >> 1), [qmax] "s"((nq - 1) * 8192),
                 [ldo32] "s"((int)(g.ldo * 64)), [cmax] "s"((nq - 1) * 128)
               : ATTN_BWD_DKV64_CLOBBERS);
#undef LO
#undef HI
}

}  // namespace

extern "C" int mgx_attn_bwd(const uint16_t* Q, const uint16_t* K, const uint16_t* V, const uint16_t* Qt, const uint16_t* Kt,
                            const uint16_t* O, const uint16_t* dO, const float* lse, float* delta, uint16_t* dOt,
                            uint16_t* dQ, uint16_t* dK, uint16_t* dV, int B, int H, int S, int Sp, long ldo, long o_bstride,
                            float scale, void* stream) {
  MGX_REQUIRE(Q && K && V && Qt && Kt && O && dO && lse && delta && dOt && dQ && dK && dV, "null operand");
  MGX_REQUIRE(B > 0 && H > 0 && S > 0 && Sp >= S && Sp % 64 == 0, "bad sizes");
  MGX_REQUIRE(ldo % 8 == 0 && o_bstride % 8 == 0, "dO rows must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  attn_bwd_prep_kernel<<<dim3(Sp / 64, H, B), 256, 0, st>>>(O, dO, ldo, o_bstride, delta, dOt, H, S, Sp);
  BwdArgs g;
  g.Q = Q; g.K = K; g.V = V; g.Qt = Qt; g.Kt = Kt; g.dO = dO; g.dOt = dOt; g.lse = lse; g.delta = delta;
  g.dQ = dQ; g.dK = dK; g.dV = dV; g.B = B; g.H = H; g.S = S; g.Sp = Sp; g.ldo = ldo; g.o_bstride = o_bstride;
  g.scale = scale; g.scale_log2e = scale * 1.4426950408889634f; g.neg_inv_scale = -1.0f / scale;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * DKV_STAGE + 65536);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * DQ_STAGE);
    attr = true;
  }
  const int nb = cdiv(S, 256) * H * B;
  const char* w64e = getenv("MGX_ATTN_W64");        // read per call: tests switch kernels inside one process
  const int w64 = w64e ? atoi(w64e) : 1;
  // the generated 64-wide kernels: S % 256 == 0, 32-bit offsets inside a 256-row block of dO, 24-bit row * ldo products
  const bool wide = w64 && S % 256 == 0 && Sp == S && ldo * 2 * 256 < (1L << 31) && ldo * 2 < (1L << 24) && Sp * 2 < (1L << 24);
  if (wide) {
    static bool attr64 = false;
    if (!attr64) {
      (void)hipFuncSetAttribute((const void*)attn_bwd_dq64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
      (void)hipFuncSetAttribute((const void*)attn_bwd_dkv64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 149504);
      attr64 = true;
    }
    attn_bwd_dkv64_kernel<<<nb, 256, 149504, st>>>(g);
    attn_bwd_dq64_kernel<<<nb, 256, 98304, st>>>(g);
  } else {
    attn_bwd_dkv_kernel<<<nb, 512, 2 * DKV_STAGE + 65536, st>>>(g);   // + wave-private V fragments
    attn_bwd_dq_kernel<<<nb, 512, 2 * DQ_STAGE, st>>>(g);
  }
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}
