// fp8 (OCP e4m3) forward of the joint attention: the "fp8 MFMA attention path" of BASELINE.json configs[4].
//
// The reference has no fp8 path (its attention is F.scaled_dot_product_attention in bf16, call sites
// fastvideo/utils/sampling_utils.py:68-82 and train_grpo_flux.py:134-144); this variant serves the same call sites
// with Q, K, V quantised per (batch, head) to e4m3 (scale 448 / amax) and both contractions on
// v_mfma_scale_f32_32x32x64_f8f6f4 (block scales fixed at 2^0): twice the bf16 MFMA rate at a quarter of the
// instructions.  Softmax statistics, row sums, the O accumulator and the LSE stay fp32; P is quantised to e4m3 in
// [0, 256].  The backward pass stays bf16 (mgx_attn_bwd with this kernel's O / LSE).
//
// Three passes:
//   mgx_attn_fp8_quantize : amax[3][B*H] of Q, K, V; Q8, K8 [B,H,S,128] e4m3; V8t [B,H,128,Sp] e4m3 with the keys of
//                           every 64-key block stored in the order the P^T accumulator feeds them to the MFMA, so the
//                           kernel's V fragments are 32 contiguous bytes per lane.
//   mgx_attn_fwd_fp8      : same structure as attn_fwd_kernel (attention.hip): 8 waves x 32 queries, 64-key tiles
//                           through two LDS buffers, S^T = K Q^T with the accumulator reused as the B operand of
//                           O^T += Vt P^T, deferred rescale.  A 64-key tile is 4 + 4 MFMAs per wave instead of 16 + 16.
#include "../../include/mixgrpo_hip.h"
#include "common.h"

#include <type_traits>

namespace {

constexpr int HD = 128;
constexpr int QW = 32;
constexpr int KB = 64;
constexpr int K8_TILE = KB * HD;   // 8 KiB
constexpr int V8_TILE = HD * KB;   // 8 KiB
constexpr float F8_MAX = 448.0f;

typedef int i32x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------ quantisation
// amax over the valid region of one tensor of one (b, h): Q / K = S rows of 128, Vt = 128 rows of S (ld Sp).
__global__ void __launch_bounds__(256) fp8_amax_kernel(const bf16_raw* __restrict__ Q, const bf16_raw* __restrict__ K,
                                                       const bf16_raw* __restrict__ Vt, uint32_t* __restrict__ amax,
                                                       int BH, int S, int Sp) {
  const int bh = blockIdx.x, t = blockIdx.y;
  const bf16_raw* p;
  int rows, cols, ld;
  if (t < 2) { p = (t == 0 ? Q : K) + (long)bh * S * HD; rows = S; cols = HD; ld = HD; }
  else { p = Vt + (long)bh * HD * Sp; rows = HD; cols = S; ld = Sp; }
  const int cpr = (cols + 7) >> 3;
  const long nvec = (long)rows * cpr;
  uint32_t m = 0;
  for (long v = (long)blockIdx.z * blockDim.x + threadIdx.x; v < nvec; v += (long)gridDim.z * blockDim.x) {
    const int row = (int)(v / cpr), c8 = (int)(v - (long)row * cpr);
    const uint4 u = *reinterpret_cast<const uint4*>(p + (long)row * ld + c8 * 8);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int col = c8 * 8 + 2 * i;
      const uint32_t lo = w[i] & 0x7FFFu, hi = (w[i] >> 16) & 0x7FFFu;   // |bf16| as an integer: order preserving
      if (col < cols) m = max(m, lo);
      if (col + 1 < cols) m = max(m, hi);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(amax + t * BH + bh, m << 16);   // fp32 bit pattern of the bf16 magnitude
}

__device__ __forceinline__ float f8_scale(float amax) { return F8_MAX / fmaxf(amax, 1e-30f); }

__device__ __forceinline__ float sat448(float x) { return fminf(fmaxf(x, -F8_MAX), F8_MAX); }

// 4 bf16 (two dwords) * sc -> 4 e4m3 bytes
__device__ __forceinline__ uint32_t quant4(uint32_t w0, uint32_t w1, float sc) {
  const float a = sat448(__builtin_bit_cast(float, w0 << 16) * sc), b = sat448(__builtin_bit_cast(float, w0 & 0xFFFF0000u) * sc);
  const float c = sat448(__builtin_bit_cast(float, w1 << 16) * sc), d = sat448(__builtin_bit_cast(float, w1 & 0xFFFF0000u) * sc);
  int r = 0;
  r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, r, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return (uint32_t)r;
}

// Q, K: same row-major layout, 16 elements per thread
__global__ void __launch_bounds__(256) fp8_quant_rows_kernel(const bf16_raw* __restrict__ Q, const bf16_raw* __restrict__ K,
                                                             uint8_t* __restrict__ Q8, uint8_t* __restrict__ K8,
                                                             const float* __restrict__ amax, int BH, long per_bh) {
  const int t = blockIdx.y;
  const bf16_raw* src = t == 0 ? Q : K;
  uint8_t* dst = t == 0 ? Q8 : K8;
  const long nvec = (long)BH * per_bh / 16;
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (long)gridDim.x * blockDim.x) {
    const long e = v * 16;
    const float sc = f8_scale(amax[t * BH + (int)(e / per_bh)]);
    const uint4 a = *reinterpret_cast<const uint4*>(src + e), b = *reinterpret_cast<const uint4*>(src + e + 8);
    uint4 o;
    o.x = quant4(a.x, a.y, sc); o.y = quant4(a.z, a.w, sc); o.z = quant4(b.x, b.y, sc); o.w = quant4(b.z, b.w, sc);
    *reinterpret_cast<uint4*>(dst + e) = o;
  }
}

// Vt -> V8t.  Position p = 32*h + 16*kb + i of a 64-key block holds key 32*kb + 8*(i >> 2) + 4*h + (i & 3): lane half h
// of the P V MFMA reads positions 32h .. 32h+31, and element 16*kb + i of its P^T fragment is accumulator register i of
// S^T block kb, whose row is exactly that key.  One thread = 16 output bytes (one kb, one h): four groups of 4 keys.
__global__ void __launch_bounds__(256) fp8_quant_vt_kernel(const bf16_raw* __restrict__ Vt, uint8_t* __restrict__ V8t,
                                                           const float* __restrict__ amax, int BH, int Sp) {
  const int q16 = Sp >> 4;                       // 16-byte output chunks per row
  const long nvec = (long)BH * HD * q16;
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (long)gridDim.x * blockDim.x) {
    const long row = v / q16;                    // (bh, d)
    const int c = (int)(v - row * q16), blk = c >> 2, qr = c & 3, h = qr >> 1, kb = qr & 1;
    const float sc = f8_scale(amax[2 * BH + (int)(row / HD)]);
    const bf16_raw* src = Vt + row * Sp + blk * 64 + kb * 32 + 4 * h;
    uint32_t o[4];
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      const uint2 u = *reinterpret_cast<const uint2*>(src + 8 * gi);
      o[gi] = quant4(u.x, u.y, sc);
    }
    *reinterpret_cast<uint4*>(V8t + row * Sp + c * 16) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// ------------------------------------------------------------------------------------------------ attention
struct AttnF8Args {
  const uint8_t* Q8;
  const uint8_t* K8;
  const uint8_t* V8t;
  const float* amax;   // [3][B*H]
  bf16_raw* O;
  float* lse;
  int B, H, S, Sp;
  long ldo, o_bstride;
  float scale_log2e;
};

// K8 tile image: [64 keys][8 chunks of 16 B], chunk ^= (key >> 1) & 7; V8t tile image: [128 d][4 chunks of 16 B],
// chunk ^= ((d >> 2) ^ (d >> 1)) & 3.  Both checked against the ds_read_b128 lane groups ({0-3,12-15,20-27},
// {4-11,16-19,28-31} per half): the 16 lanes of a group read the same chunk index of 16 different rows and land on 16
// different bank quads.
__device__ __forceinline__ int k8_off(int key, int chunk) { return key * 128 + ((chunk ^ ((key >> 1) & 7)) << 4); }
__device__ __forceinline__ int v8_off(int d, int chunk) { return d * 64 + ((chunk ^ (((d >> 2) ^ (d >> 1)) & 3)) << 4); }

__device__ __forceinline__ i32x8 frag32(const char* p0, const char* p1) {
  const uint4 a = *reinterpret_cast<const uint4*>(p0), b = *reinterpret_cast<const uint4*>(p1);
  i32x8 f;
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  return f;
}

#define MFMA_F8(A_, B_, C_) \
  __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A_, B_, C_, 0, 0, 0, one_e8m0, 0, one_e8m0)

template <bool DEFER>
__global__ void __launch_bounds__(512, 2) attn_fwd_fp8_kernel(AttnF8Args g) {
  constexpr int NW = 8, QB = QW * NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][K8 tile | V8t tile]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  int one_e8m0 = 0x7F7F7F7F;                                     // E8M0 block scale 2^0 in every byte
  asm volatile("" : "+v"(one_e8m0));

  const int nq = (g.S + QB - 1) / QB;
  const int nwg = nq * g.H * g.B;
  int bid = blockIdx.x;
  {   // the q-tiles of one (b, head) stay on one XCD so its K/V stay in that L2
    const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
  }
  const int qt = bid % nq;
  const int bh = bid / nq;
  const int b = bh / g.H, hh = bh - b * g.H;
  const int BH = g.B * g.H;
  const float aq = g.amax[bh], ak = g.amax[BH + bh], av = g.amax[2 * BH + bh];
  // stored value = x * 448 / amax, so a raw score is q.k * (448/aq) * (448/ak)
  // (floored: an all-zero Q or K must give c > 0 so that (mx - -inf) * c is +inf, not NaN)
  const float c = fmaxf(g.scale_log2e * (fmaxf(aq, 1e-30f) * (1.0f / F8_MAX)) * (fmaxf(ak, 1e-30f) * (1.0f / F8_MAX)), 1e-30f);

  const uint8_t* Qp = g.Q8 + (long)bh * g.S * HD;
  const uint8_t* Kp = g.K8 + (long)bh * g.S * HD;
  const uint8_t* Vp = g.V8t + (long)bh * HD * g.Sp;

  const int q0 = qt * QB + wid * QW;
  int qrow = q0 + r;
  if (qrow >= g.S) qrow = g.S - 1;
  i32x8 qf[2];   // B operand of S^T = K Q^T: Q8[q][64*ks + 32*h + j]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const char* p = reinterpret_cast<const char*>(Qp + (long)qrow * HD + ks * 64 + h * 32);
    qf[ks] = frag32(p, p + 16);
  }

  // staging: one 16-byte chunk of the K8 tile and one of the V8t tile per thread
  const int kc_key = tid >> 3, kc_chunk = tid & 7;
  const int vc_d = tid >> 2, vc_chunk = tid & 3;
  uint4 sk, sv;
  const int ntiles = (g.S + KB - 1) / KB;
#define LOAD_KV(t)                                                                               \
  do {                                                                                           \
    int ka = (t) * KB + kc_key;                                                                  \
    if (ka >= g.S) ka = g.S - 1;                                                                 \
    sk = *reinterpret_cast<const uint4*>(Kp + (long)ka * HD + kc_chunk * 16);                    \
    sv = *reinterpret_cast<const uint4*>(Vp + (long)vc_d * g.Sp + (t) * KB + vc_chunk * 16);     \
  } while (0)
#define STORE_KV(buf)                                                                            \
  do {                                                                                           \
    char* kb_ptr = smem + (buf) * (K8_TILE + V8_TILE);                                           \
    *reinterpret_cast<uint4*>(kb_ptr + k8_off(kc_key, kc_chunk)) = sk;                           \
    *reinterpret_cast<uint4*>(kb_ptr + K8_TILE + v8_off(vc_d, vc_chunk)) = sv;                   \
  } while (0)

  f32x16 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  LOAD_KV(0);
  STORE_KV(0);
  __syncthreads();
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) asm volatile("" ::"v"(qf[ks]));
  int cur = 0;
  auto tile = [&](int t, auto mask_tag) __attribute__((always_inline)) {
    constexpr bool MASK = decltype(mask_tag)::value;
    if (t + 1 < ntiles) LOAD_KV(t + 1);
    const char* ks_ = smem + cur * (K8_TILE + V8_TILE);
    const char* vs_ = ks_ + K8_TILE;

    // ---- S^T blocks (32 keys x 32 queries) x 2: two K = 64 MFMAs each
    i32x8 kf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        kf[kb][ks] = frag32(ks_ + k8_off(kb * 32 + r, 4 * ks + 2 * h), ks_ + k8_off(kb * 32 + r, 4 * ks + 2 * h + 1));
    __builtin_amdgcn_sched_barrier(0);
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
      s[kb] = MFMA_F8(kf[kb][0], qf[0], s[kb]);
      s[kb] = MFMA_F8(kf[kb][1], qf[1], s[kb]);
    }
    // V fragments: requested here so that they return underneath the softmax
    i32x8 vf[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vf[dt] = frag32(vs_ + v8_off(dt * 32 + r, 2 * h), vs_ + v8_off(dt * 32 + r, 2 * h + 1));
    __builtin_amdgcn_sched_barrier(0);
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
    // ---- online softmax, deferred rescale as in attn_fwd_kernel.  P = 4 * 2^(s*c - m*c): at most 4 * 2^6 = 256 < 448
    float mx = s[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[kb][i]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    constexpr float DEFER_LOG2 = 6.0f;
    bool rescale = true;
    if (DEFER) rescale = __builtin_amdgcn_ballot_w64((mx - m_run) * c > DEFER_LOG2) != 0;
    float alpha = 1.0f;
    if (rescale) {
      const float m_new = fmaxf(m_run, mx);
      alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      m_run = m_new;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
    }
    const float mc = m_run * c - 2.0f;
    float psum = 0.f;
    i32x8 pf;   // P^T fragment: element 16*kb + i = accumulator register i of block kb
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        const float p0 = __builtin_amdgcn_exp2f(s[kb][4 * i4] * c - mc);
        const float p1 = __builtin_amdgcn_exp2f(s[kb][4 * i4 + 1] * c - mc);
        const float p2 = __builtin_amdgcn_exp2f(s[kb][4 * i4 + 2] * c - mc);
        const float p3 = __builtin_amdgcn_exp2f(s[kb][4 * i4 + 3] * c - mc);
        psum += (p0 + p1) + (p2 + p3);
        int wv = 0;
        wv = __builtin_amdgcn_cvt_pk_fp8_f32(p0, p1, wv, false);
        wv = __builtin_amdgcn_cvt_pk_fp8_f32(p2, p3, wv, true);
        pf[kb * 4 + i4] = wv;
      }
    l_run = l_run * alpha + psum;
    // ---- O^T += V8t P^T: the whole 64-key tile is one K = 64 step per 32-row d tile
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = MFMA_F8(vf[dt], pf, o[dt]);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < ntiles) STORE_KV(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  };
  const int nfull = g.S / KB;
  for (int t = 0; t < nfull; ++t) tile(t, std::false_type{});
  if (nfull < ntiles) tile(nfull, std::true_type{});

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = (1.0f / l_tot) * (av * (1.0f / F8_MAX));     // V8 = V * 448 / av
  const int q = q0 + r;
  if (q < g.S) {
    bf16_raw* op = g.O + (long)b * g.o_bstride + (long)q * g.ldo + hh * HD;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        const int d = dt * 32 + 8 * i4 + 4 * h;
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        const f32x2_t v0 = {o[dt][4 * i4] * inv, o[dt][4 * i4 + 1] * inv};
        const f32x2_t v1 = {o[dt][4 * i4 + 2] * inv, o[dt][4 * i4 + 3] * inv};
        uint2 w;
        w.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(v0, bf16x2_t));
        w.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(v1, bf16x2_t));
        *reinterpret_cast<uint2*>(op + d) = w;
      }
    if (g.lse && h == 0)   // sum_k 2^(s c) = 2^(m c - 2) * l
      g.lse[(long)bh * g.S + q] = (m_run * c - 2.0f) * 0.6931471805599453f + logf(l_tot);
  }
}

}  // namespace

extern "C" int mgx_attn_fp8_quantize(const uint16_t* Q, const uint16_t* K, const uint16_t* Vt, uint8_t* Q8, uint8_t* K8,
                                     uint8_t* V8t, float* amax, int B, int H, int S, int Sp, void* stream) {
  MGX_REQUIRE(Q && K && Vt && Q8 && K8 && V8t && amax, "null operand");
  MGX_REQUIRE(B > 0 && H > 0 && S > 0, "empty attention");
  MGX_REQUIRE(Sp >= S && Sp % 64 == 0, "Sp must be S rounded up to a multiple of 64");
  hipStream_t st = (hipStream_t)stream;
  const int BH = B * H;
  if (hipMemsetAsync(amax, 0, sizeof(float) * 3 * BH, st) != hipSuccess) {
    mgx_set_error("mgx_attn_fp8_quantize: hipMemsetAsync failed");
    return MGX_ERR_LAUNCH;
  }
  const int zc = max(1, min(64, cdiv((long)S * HD / 8, 256 * 8)));
  fp8_amax_kernel<<<dim3(BH, 3, zc), 256, 0, st>>>(Q, K, Vt, reinterpret_cast<uint32_t*>(amax), BH, S, Sp);
  const long per_bh = (long)S * HD;
  fp8_quant_rows_kernel<<<dim3(min(cdiv((long)BH * per_bh / 16, 256), 16384), 2), 256, 0, st>>>(Q, K, Q8, K8, amax, BH,
                                                                                                  per_bh);
  fp8_quant_vt_kernel<<<min(cdiv((long)BH * HD * (Sp / 16), 256), 16384), 256, 0, st>>>(Vt, V8t, amax, BH, Sp);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}

extern "C" int mgx_attn_fwd_fp8(const uint8_t* Q8, const uint8_t* K8, const uint8_t* V8t, const float* amax, uint16_t* O,
                                float* lse, int B, int H, int S, int Sp, long ldo, long o_bstride, float scale,
                                void* stream) {
  MGX_REQUIRE(Q8 && K8 && V8t && amax && O, "null operand");
  MGX_REQUIRE(B > 0 && H > 0 && S > 0, "empty attention");
  MGX_REQUIRE(Sp >= S && Sp % 64 == 0, "Sp must be S rounded up to a multiple of 64");
  MGX_REQUIRE(ldo % 4 == 0 && o_bstride % 4 == 0, "output strides must keep 8-byte alignment");
  MGX_REQUIRE(scale >= 0.f, "the running-maximum logic needs a non-negative scale");
  AttnF8Args g;
  g.Q8 = Q8; g.K8 = K8; g.V8t = V8t; g.amax = amax; g.O = O; g.lse = lse;
  g.B = B; g.H = H; g.S = S; g.Sp = Sp; g.ldo = ldo; g.o_bstride = o_bstride;
  g.scale_log2e = scale * 1.4426950408889634f;
  const int lds = 2 * (K8_TILE + V8_TILE);
  attn_fwd_fp8_kernel<true><<<cdiv(S, 256) * H * B, 512, lds, (hipStream_t)stream>>>(g);
  MGX_CHECK_LAUNCH();
  return MGX_OK;
}
