// Joint text+image attention for the FLUX MMDiT (non-causal, no mask, head_dim 128) -- forward.
//
// Replaces F.scaled_dot_product_attention as called by diffusers' FluxAttnProcessor2_0 under autocast(bf16)
// (reference call sites fastvideo/utils/sampling_utils.py:68-82, train_grpo_flux.py:134-144).
//
// Layout: Q, K [B, H, S, 128] bf16 (RMS-normed + RoPE'd by qk_norm_rope), V transposed Vt [B, H, 128, Sp] bf16
// (Sp = S rounded up to 64, padding finite).  Output O [B, S, ldo] bf16 at column h*128 (ldo = d or 5d).
//
// Structure (CDNA4, wave64): one workgroup = 8 waves = 256 query rows of one (b, h); each wave owns 32 queries.
// K/V tiles of 64 keys go global -> registers -> LDS (issue-early / write-late, two LDS buffers, one barrier per
// tile).  S^T = K Q^T on v_mfma_f32_32x32x16_bf16 with K as the A operand, so a lane holds one query column
// and 16 of the 32 keys of a block: the softmax row statistics are lane-local plus ONE exchange with lane^32.
// The S^T accumulator is converted in place to the B operand of O^T += Vt P^T (accumulator-as-operand, no LDS
// round trip); Vt is stored in LDS in the matching permuted key order, so both the K and the Vt fragments are single,
// bank-conflict-free ds_read_b128 (XOR-swizzled images).
#include "../../include/mixgrpo_hip.h"
#include "common.h"

#include <cstdlib>
#include <type_traits>

namespace {

constexpr int HD = 128;
constexpr int QW = 32;            // queries per wave
constexpr int KB = 64;            // keys per tile
constexpr int K_TILE_BYTES = KB * HD * 2;   // 16 KiB
constexpr int V_TILE_BYTES = HD * KB * 2;   // 16 KiB

struct AttnArgs {
  const bf16_raw* Q;
  const bf16_raw* K;
  const bf16_raw* Vt;
  bf16_raw* O;
  float* lse;      // [B, H, S] natural-log LSE of scale*scores (for the backward), may be null
  int B, H, S, Sp;
  long ldo;        // elements between consecutive tokens of O
  long o_bstride;  // elements between batches of O
  float scale_log2e;
};

// K tile image: [64 keys][16 chunks of 16 B], chunk ^= key & 15
__device__ __forceinline__ int k_off(int key, int chunk) { return key * 256 + ((chunk ^ (key & 15)) << 4); }
// Vt tile image: [128 d][8 slots of 16 B].  Slot (b, h) of a row holds, for the 16-key block b, the keys 4h+{0..3} and
// 4h+8+{0..3}: exactly the 8 keys (in the S^T accumulator's order) that lane half h feeds to one P V MFMA, so the A
// operand is ONE ds_read_b128.  (Two ds_read_b64 per operand get fused by hipcc into ds_read2st64_b64, which runs at
// half the LDS rate and banks modulo 32: the 64-bank swizzle then conflicts 2-way -- SQ_LDS_BANK_CONFLICT was 39 % of
// the LDS-active cycles and the LDS port, not the matrix pipe, paced the kernel.)  slot ^= (d >> 1) & 7.
__device__ __forceinline__ int v_off(int d, int slot16) { return d * 128 + ((slot16 ^ ((d >> 1) & 7)) << 4); }

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));   // one v_cvt_pk_bf16_f32
}

// NW waves per workgroup (QB = 32*NW queries).  NW = 4 puts two independent workgroups on a CU (64 KiB LDS each): their
// barriers are unrelated, so one workgroup's softmax VALU phase runs under the other's MFMA phase instead of the two
// waves of a SIMD marching in lockstep.
// OVL: the P V product of the tile's first 32 keys (8 MFMAs) is issued BESIDE the exponentials of its last 32 keys, inside
// one scheduling region with `sched_group_barrier` hints (1 MFMA : 2 v_exp : 5 other VALU per gap -- 36 issue cycles per
// 32-cycle MFMA).  In-kernel stamps showed the softmax VALU phase (1100-1600 cycles per wave and tile) and the MFMA phases
// (2 x ~600) adding up almost serially: both waves of a SIMD are in the same phase at the same time.
template <int NW, bool DEFER, bool OVL = false>
__global__ void __launch_bounds__(NW * 64, 2) attn_fwd_kernel(AttnArgs g) {
  constexpr int QB = QW * NW;
  constexpr int NCH = 1024 / (NW * 64);   // 16-byte chunks per thread per operand tile
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][K tile | Vt tile]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  const int nq = (g.S + QB - 1) / QB;
  // XCD-aware order: the q-tiles of one (b, head) stay on one XCD so its K/V stay in that L2
  const int nwg = nq * g.H * g.B;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
  }
  const int qt = bid % nq;
  const int bh = bid / nq;
  const int b = bh / g.H, hh = bh - b * g.H;

  const bf16_raw* Qp = g.Q + (long)bh * g.S * HD;
  const bf16_raw* Kp = g.K + (long)bh * g.S * HD;
  const bf16_raw* Vp = g.Vt + (long)bh * HD * g.Sp;

  // ---- this wave's Q fragments (B operand of S^T = K Q^T): Q[q0 + r][16*ks + 8*h + j]
  const int q0 = qt * QB + wid * QW;
  int qrow = q0 + r;
  if (qrow >= g.S) qrow = g.S - 1;
  s16x8 qf[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
    qf[ks] = *reinterpret_cast<const s16x8*>(Qp + (long)qrow * HD + ks * 16 + h * 8);

  // ---- staging: K tile = 1024 16-B chunks, Vt tile = 1024 16-B chunks; NCH of each per thread
  const int kc_key0 = tid >> 4, kc_chunk = tid & 15;          // chunk id = tid + NW*64*i: key = id>>4
  const int vc_d0 = tid >> 3, vc_chunk = tid & 7;             // chunk id = tid + NW*64*i: d = id>>3, 16-B chunk of 64 keys
  constexpr int KSTEP = NW * 4, DSTEP = NW * 8;               // keys / d rows advanced per pass
  uint4 sk0, sk1, sk2, sk3, sv0, sv1, sv2, sv3;   // named (not an array): keeps the staging registers out of scratch
  const int ntiles = (g.S + KB - 1) / KB;
#define LOAD1(i_, SK, SV)                                                                           \
  if constexpr (NCH > i_) {                                                                         \
    int ka = key_base + kc_key0 + KSTEP * i_;                                                       \
    if (ka >= g.S) ka = g.S - 1;                                                                    \
    SK = *reinterpret_cast<const uint4*>(Kp + (long)ka * HD + kc_chunk * 8);                        \
    SV = *reinterpret_cast<const uint4*>(Vp + (long)(vc_d0 + DSTEP * i_) * g.Sp + key_base + vc_chunk * 8); \
  }
#define LOAD_KV(t)                                                                                  \
  do {                                                                                              \
    const int key_base = (t) * KB;                                                                  \
    LOAD1(0, sk0, sv0) LOAD1(1, sk1, sv1) LOAD1(2, sk2, sv2) LOAD1(3, sk3, sv3)                     \
  } while (0)
#define STORE1(i_, SK, SV)                                                                          \
  if constexpr (NCH > i_) {                                                                         \
    *reinterpret_cast<uint4*>(kb_ptr + k_off(kc_key0 + KSTEP * i_, kc_chunk)) = SK;                 \
    /* 16-byte chunk vc_chunk = keys 8c..8c+7 of the tile: 4-key groups g = 2c, 2c+1 of block b = c >> 1 */ \
    *reinterpret_cast<uint2*>(vb_ptr + v_off(vc_d0 + DSTEP * i_, (vc_chunk >> 1) * 2) + (vc_chunk & 1) * 8) = make_uint2(SV.x, SV.y); \
    *reinterpret_cast<uint2*>(vb_ptr + v_off(vc_d0 + DSTEP * i_, (vc_chunk >> 1) * 2 + 1) + (vc_chunk & 1) * 8) = make_uint2(SV.z, SV.w); \
  }
#define STORE_KV(buf)                                                                               \
  do {                                                                                              \
    char* kb_ptr = smem + (buf) * (K_TILE_BYTES + V_TILE_BYTES);                                    \
    char* vb_ptr = kb_ptr + K_TILE_BYTES;                                                           \
    STORE1(0, sk0, sv0) STORE1(1, sk1, sv1) STORE1(2, sk2, sv2) STORE1(3, sk3, sv3)                 \
  } while (0)

  f32x16 o[4];   // O^T tiles: d in [32*dt, 32*dt+32), column = query r
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
  float m_run = -INFINITY;   // running max of raw scores (shared by both half-waves)
  float l_run = 0.f;         // this lane's partial row sum

  LOAD_KV(0);
  STORE_KV(0);
  __syncthreads();
  // make the Q-fragment loads provably complete before the loop: otherwise hipcc's waitcnt pass carries them as
  // pending into the loop and fences every S^T MFMA behind the (slow) K/V staging loads of the same iteration
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) asm volatile("" :: "v"(qf[ks]));
  int cur = 0;
  // one K/V tile: S^T, online softmax, O^T accumulation.  Kept as a lambda over a compile-time MASK so the full
  // tiles carry no masking selects (hipcc if-converts a runtime "last tile" test into ~200 v_cndmask per tile).
#define STAMP(k_) do {} while (0)
  auto tile = [&](int t, auto mask_tag) __attribute__((always_inline)) {
    constexpr bool MASK = decltype(mask_tag)::value;
    if (t + 1 < ntiles) LOAD_KV(t + 1);
    STAMP(0);
    const char* ks_ = smem + cur * (K_TILE_BYTES + V_TILE_BYTES);
    const char* vs_ = ks_ + K_TILE_BYTES;

    // ---- S^T blocks (32 keys x 32 queries) x 2.  All eight K fragments of a block are requested before its MFMA
    // chain, and the second block's fragments re-fill each register as soon as its MFMA has issued: written as
    // "read; mfma" pairs, hipcc funnels every fragment through ONE register quad and emits read -> lgkmcnt(0) -> MFMA
    // sixteen times, exposing the LDS latency on every MFMA.
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
    {
      s16x8 kf[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) kf[ks] = *reinterpret_cast<const s16x8*>(ks_ + k_off(r, ks * 2 + h));
      __builtin_amdgcn_sched_barrier(0);           // (the scheduler otherwise sinks every read back to its MFMA)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s[0], 0, 0, 0);
        kf[ks] = *reinterpret_cast<const s16x8*>(ks_ + k_off(32 + r, ks * 2 + h));
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s[1], 0, 0, 0);
    }
    STAMP(1);
    // first Vt fragments of the P V product: requested here so that they return underneath the softmax
    s16x8 vfr[2][4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vfr[0][dt] = *reinterpret_cast<const s16x8*>(vs_ + v_off(dt * 32 + r, h));
    __builtin_amdgcn_sched_barrier(0);
    // ---- mask keys beyond S (only the ragged last tile is compiled with MASK)
    if (MASK) {
      const int key_base = t * KB;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = key_base + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (key >= g.S) s[kb][i] = -INFINITY;
        }
    }
    // ---- online softmax (lane = one query; 32 of the tile's 64 keys here, 32 in lane^32)
    float mx = s[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[kb][i]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    // Deferred rescale (DEFER): the running maximum is only raised -- and O, l rescaled -- when some row of the wave
    // has outgrown it by more than 2^DEFER_LOG2; otherwise the stale maximum is kept and this tile's P may reach
    // 2^DEFER_LOG2 instead of 1 (fp32 row sums and accumulators, bf16 P: the relative precision is unchanged).  The
    // decision is taken BEFORE this tile's P is exponentiated and after the previous tile's P V has been issued, so
    // everything still at the old scale (O, l) is rescaled exactly once and nothing at the new scale is.  It saves the
    // 64 v_mul of the O rescale (a quarter of the tile's VALU work) on almost every tile.
    constexpr float DEFER_LOG2 = 6.0f;
    bool rescale = true;
    if (DEFER) rescale = __builtin_amdgcn_ballot_w64((mx - m_run) * g.scale_log2e > DEFER_LOG2) != 0;   // -inf run max: true
    float alpha = 1.0f;
    if (rescale) {
      const float m_new = fmaxf(m_run, mx);
      alpha = __builtin_amdgcn_exp2f((m_run - m_new) * g.scale_log2e);   // m_run = -inf on the first tile -> 0
      m_run = m_new;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
    }
    const float mc = m_run * g.scale_log2e;
    float psum = 0.f;
    uint32_t pb[2][8];   // P^T as bf16 pairs: B-operand fragments, k-step s uses regs 8s..8s+7
    auto exps = [&](int kb) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        const float p0 = __builtin_amdgcn_exp2f(s[kb][i] * g.scale_log2e - mc);
        const float p1 = __builtin_amdgcn_exp2f(s[kb][i + 1] * g.scale_log2e - mc);
        psum += p0 + p1;
        pb[kb][i >> 1] = pack_bf16(p0, p1);
      }
    };
    auto pfrag = [&](int idx) __attribute__((always_inline)) {
      const int kb = idx >> 1, s2 = idx & 1;
      return __builtin_bit_cast(s16x8, make_uint4(pb[kb][4 * s2], pb[kb][4 * s2 + 1], pb[kb][4 * s2 + 2], pb[kb][4 * s2 + 3]));
    };
    if constexpr (OVL) {
      exps(0);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) vfr[1][dt] = *reinterpret_cast<const s16x8*>(vs_ + v_off(dt * 32 + r, 2 + h));
      __builtin_amdgcn_sched_barrier(0);
      // ---- one scheduling region: 8 MFMAs (keys 0..31) + the exponentials of keys 32..63 + the V fragments of idx 2
      {
        const s16x8 pf0 = pfrag(0), pf1 = pfrag(1);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[0][dt], pf0, o[dt], 0, 0, 0);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) vfr[0][dt] = *reinterpret_cast<const s16x8*>(vs_ + v_off(dt * 32 + r, 4 + h));
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[1][dt], pf1, o[dt], 0, 0, 0);
        exps(1);
#pragma unroll
        for (int k_ = 0; k_ < 8; ++k_) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
          __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);     // 2 transcendental (v_exp_f32)
          __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);     // 5 other VALU
          if (k_ == 3) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // the four V fragment reads of idx 2
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      l_run = l_run * alpha + psum;
      STAMP(2);
#pragma unroll
      for (int idx = 2; idx < 4; ++idx) {
        const s16x8 pf = pfrag(idx);
        if (idx + 1 < 4) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
            vfr[(idx + 1) & 1][dt] = *reinterpret_cast<const s16x8*>(vs_ + v_off(dt * 32 + r, (idx + 1) * 2 + h));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[idx & 1][dt], pf, o[dt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
    exps(0);
    exps(1);
    l_run = l_run * alpha + psum;
    STAMP(2);

    // ---- O^T += Vt P^T : A = Vt[d = 32*dt + r][slot (b = 2 kb + s2, h)]: the 8 keys of lane half h in the S^T
    // accumulator's order; the four fragments of step idx + 1 are requested before the MFMAs of step idx
#pragma unroll
    for (int idx = 0; idx < 4; ++idx) {
      const s16x8 pf = pfrag(idx);
      if (idx + 1 < 4) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          vfr[(idx + 1) & 1][dt] = *reinterpret_cast<const s16x8*>(vs_ + v_off(dt * 32 + r, (idx + 1) * 2 + h));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[idx & 1][dt], pf, o[dt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    }
    STAMP(3);
    if (t + 1 < ntiles) STORE_KV(cur ^ 1);
    STAMP(4);
    __syncthreads();
    STAMP(5);
    cur ^= 1;
  };
  const int nfull = g.S / KB;
  for (int t = 0; t < nfull; ++t) tile(t, std::false_type{});
  if (nfull < ntiles) tile(nfull, std::true_type{});

#undef STAMP
  // ---- finalize: row sum across the two half-waves, normalise, store O[q][h*128 + d]
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int q = q0 + r;
  if (q < g.S) {
    bf16_raw* op = g.O + (long)b * g.o_bstride + (long)q * g.ldo + hh * HD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        const int d = dt * 32 + 8 * i4 + 4 * h;
        uint2 w;
        w.x = pack_bf16(o[dt][4 * i4] * inv, o[dt][4 * i4 + 1] * inv);
        w.y = pack_bf16(o[dt][4 * i4 + 2] * inv, o[dt][4 * i4 + 3] * inv);
        *reinterpret_cast<uint2*>(op + d) = w;
      }
    if (g.lse && h == 0) {
      // ln sum_k exp(scale*s_k) = m*scale + ln(l)
      g.lse[(long)bh * g.S + q] = m_run * (g.scale_log2e * 0.6931471805599453f) + logf(l_tot);
    }
  }
}

// ------------------------------------------------------------------------------------------------ 64-query waves
// attn_fwd64_kernel: 4 waves x 64 queries (two 32-query chains that share every K / V^T fragment), one wave per SIMD with
// all 512 registers, K / V^T tiles by LDS-DMA.  The body is a generated, hand-placed instruction stream: design, register
// map and schedule in csrc/gen/attn_fwd64.py; checked on the CPU by tests/asm_emu.py (interpreter + hazard pass) before it
// runs here.  Needs S % 256 == 0 (FLUX: 512 + 4096 tokens); other shapes take attn_fwd_kernel above.
// ACC (attn_fwd64q_body.inc, mgx_attn_fwd_log2): Q arrives multiplied by scale * log2(e), the scores are exponents of two, and
// from a block's second tile on the MFMA accumulator of a score tile starts at -m, so the softmax is v_exp_f32 straight from the
// accumulator (no multiply-add per score: the loop is VALU-issue-bound, gen/attn_fwd64.py).
#include "attn_fwd64_body.inc"
#include "attn_fwd64q_body.inc"

#define ATTN_FWD64_OPERANDS \
  [tid] "v"(threadIdx.x), [q_lo] "s"((unsigned)qp), [q_hi] "s"((unsigned)(qp >> 32)), [k_lo] "s"((unsigned)kp), \
                 [k_hi] "s"((unsigned)(kp >> 32)), [v_lo] "s"((unsigned)vp), [v_hi] "s"((unsigned)(vp >> 32)), \
                 [o_lo] "s"((unsigned)op), [o_hi] "s"((unsigned)(op >> 32)), [l_lo] "s"((unsigned)lp), \
                 [l_hi] "s"((unsigned)(lp >> 32)), [sp2] "s"(g.Sp * 2), [ldo2] "s"((int)(g.ldo * 2)), [cs] "s"(g.scale_log2e), \
                 [nloop] "s"((ntiles - 2) >> 1), [kmax] "s"((ntiles - 1) * 16384), [vmax] "s"((ntiles - 1) * 128), \
                 [nblk] "s"(count), [qt0] "s"(qt), [hh0] "s"(hh), [b0] "s"(b), [nq] "s"(nq), [nh] "s"(g.H), [kstep] "s"(g.S * 256), \
                 [ostep] "s"(ostep), [obs] "s"((int)(g.o_bstride * 2)), [ob_lo] "s"((unsigned)ob), [ob_hi] "s"((unsigned)(ob >> 32)), \
                 [sq] "s"(stride % nq), [dbh] "s"(stride / nq), [qstride] "s"(stride * 65536), [lstride] "s"(stride * 1024)

template <bool ACC>
__global__ void __launch_bounds__(256, 1) attn_fwd64_kernel(AttnArgs g) {
  // Persistent: the blocks (batch, head, q-tile) are cut into 8 contiguous ranges, one per XCD (blockIdx & 7 under round-robin
  // dispatch: speed only); workgroup j of an XCD takes blocks lo + j, lo + j + stride, ...  (stride = workgroups per XCD), so the
  // XCD's workgroups work on neighbouring q-tiles of the same heads at the same time and share their K / V in L2.  The next
  // block's first tiles and Q fragments are fetched during the current block's last iteration.
  const int nq = g.S >> 8;
  const long nblk = (long)nq * g.H * g.B;
  const int G = gridDim.x, w = blockIdx.x;
  int first, count, stride;
  if ((G & 7) == 0) {
    const int x = w & 7, j = w >> 3;
    stride = G >> 3;
    const long lo = x * nblk / 8, hi = (x + 1) * nblk / 8;
    first = (int)(lo + j);
    count = first < hi ? (int)((hi - first + stride - 1) / stride) : 0;
  } else {
    stride = G;
    first = w;
    count = first < nblk ? (int)((nblk - first + stride - 1) / stride) : 0;
  }
  if (count <= 0) return;
  const int qt = first % nq;
  const int bh = first / nq;
  const int b = bh / g.H, hh = bh - b * g.H;
  const unsigned long long qp = (unsigned long long)(g.Q + ((long)bh * g.S + qt * 256) * HD);
  const unsigned long long kp = (unsigned long long)(g.K + (long)bh * g.S * HD);
  const unsigned long long vp = (unsigned long long)(g.Vt + (long)bh * HD * g.Sp);
  const unsigned long long ob = (unsigned long long)g.O;
  const unsigned long long op = (unsigned long long)(g.O + (long)b * g.o_bstride + (long)(qt * 256) * g.ldo + hh * HD);
  const unsigned long long lp = g.lse ? (unsigned long long)(g.lse + (long)bh * g.S + qt * 256) : 0ull;
  const int ntiles = g.S >> 6;
  const int ostep = (int)(g.ldo * 512);                       // bytes of 256 rows of O
  if constexpr (ACC) asm volatile(ATTN_FWD64Q_BODY : : ATTN_FWD64_OPERANDS : ATTN_FWD64Q_CLOBBERS);
  else asm volatile(ATTN_FWD64_BODY : : ATTN_FWD64_OPERANDS : ATTN_FWD64_CLOBBERS);
}

}  // namespace

static int attn_fwd_any(const uint16_t* Q, const uint16_t* K, const uint16_t* Vt, uint16_t* O, float* lse, int B, int H, int S,
                        int Sp, long ldo, long o_bstride, float scale_log2e, bool log2_scores, void* stream) {
  MGX_REQUIRE(Q && K && Vt && O, "null operand");
  MGX_REQUIRE(B > 0 && H > 0 && S > 0, "empty attention");
  MGX_REQUIRE(Sp >= S && Sp % 64 == 0, "Sp must be S rounded up to a multiple of 64");
  MGX_REQUIRE(ldo % 4 == 0 && o_bstride % 4 == 0, "output strides must keep 8-byte alignment");
  AttnArgs g;
  g.Q = Q; g.K = K; g.Vt = Vt; g.O = O; g.lse = lse;
  g.B = B; g.H = H; g.S = S; g.Sp = Sp; g.ldo = ldo; g.o_bstride = o_bstride;
  g.scale_log2e = scale_log2e;
  static const int nw = getenv("MGX_ATTN_NW") ? atoi(getenv("MGX_ATTN_NW")) : 8;
  static const int defer = getenv("MGX_ATTN_DEFER") ? atoi(getenv("MGX_ATTN_DEFER")) : 1;
  const int lds = 2 * (K_TILE_BYTES + V_TILE_BYTES);
  hipStream_t st = (hipStream_t)stream;
  static const int ovl = getenv("MGX_ATTN_OVL") ? atoi(getenv("MGX_ATTN_OVL")) : 1;
  const char* w64e = getenv("MGX_ATTN_W64");   // read per call: tests switch kernels inside one process
  const int w64 = w64e ? atoi(w64e) : 1;
  // The persistent walk advances (q-tile, head, batch) by a fixed stride and carries ONCE per step (gen/attn_fwd64.py,
  // block_advance_stores: head -= H, batch += 1), so a step must move the head index by less than H: stride / nq < H.
  // (FLUX: H = 24, nq >= 3, stride 32.)  Shapes outside that -- S = 256 with many batches, few heads -- take attn_fwd_kernel.
  const long nblk64 = (long)(S / 256) * H * B;
  const int grid64 = nblk64 >= 256 ? 256 : (int)nblk64;
  const int stride64 = (grid64 & 7) == 0 ? grid64 >> 3 : grid64;
  const bool walk_ok = S >= 256 && stride64 / (S / 256) < H;
  if (w64 && walk_ok && S % 256 == 0 && Sp == S && (long)S * ldo * 2 < (1L << 31) && o_bstride * 2 < (1L << 31) &&
      (long)S * 256 < (1L << 31)) {
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)attn_fwd64_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
      (void)hipFuncSetAttribute((const void*)attn_fwd64_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
      attr = true;
    }
    if (log2_scores) attn_fwd64_kernel<true><<<grid64, 256, 65536, st>>>(g);
    else attn_fwd64_kernel<false><<<grid64, 256, 65536, st>>>(g);
    MGX_CHECK_LAUNCH();
    return MGX_OK;
  }
  if (nw == 8 && defer && ovl) attn_fwd_kernel<8, true, true><<<cdiv(S, 256) * H * B, 512, lds, st>>>(g);
  else if (nw == 8 && defer) attn_fwd_kernel<8, true><<<cdiv(S, 256) * H * B, 512, lds, st>>>(g);
  else if (nw == 8) attn_fwd_kernel<8, false><<<cdiv(S, 256) * H * B, 512, lds, st>>>(g);
  else if (defer) attn_fwd_kernel<4, true><<<cdiv(S, 128) * H * B, 256, lds, st>>>(g);
  else attn_fwd_kernel<4, false><<<cdiv(S, 128) * H * B, 256, lds, st>>>(g);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int mgx_attn_fwd(const uint16_t* Q, const uint16_t* K, const uint16_t* Vt, uint16_t* O, float* lse, int B,
                            int H, int S, int Sp, long ldo, long o_bstride, float scale, void* stream) {
  return attn_fwd_any(Q, K, Vt, O, lse, B, H, S, Sp, ldo, o_bstride, scale * 1.4426950408889634f, false, stream);
}

extern "C" int mgx_attn_fwd_log2(const uint16_t* Q2, const uint16_t* K, const uint16_t* Vt, uint16_t* O, float* lse, int B,
                                 int H, int S, int Sp, long ldo, long o_bstride, void* stream) {
  return attn_fwd_any(Q2, K, Vt, O, lse, B, H, S, Sp, ldo, o_bstride, 1.0f, true, stream);
}
