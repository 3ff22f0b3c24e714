// SCRATCH A/B (not product): gemm_pp_kernel's ping-pong K-loop with the two bf16 MFMA shapes, everything else equal.
//   arm 16: v_mfma_f32_16x16x32_bf16, the shipped loop (64 MFMAs per wave and K-tile, 24 ds_read_b128)
//   arm 32: v_mfma_f32_32x32x16_bf16 (32 MFMAs of twice the duration, the same 24 ds_read_b128: half the operand-register
//           reads per FLOP -- profiles/r03_mfma_power.log: +3.4 % at the chip's limiter in an MFMA-only loop)
// Same 256 x 256 x 64 tiles, LDS images (3 A + 2 W stages, XOR swizzle through the DMA source address), LDS-DMA pieces, barrier
// protocol, persistent XCD-aware tile walk.  Plain matrices, M, N % 256 == 0, K % 64 == 0, K >= 128; C = bf16(A W^T), no bias.
// Epilogues: each arm stores straight from registers with 16-byte accesses; the W rows of a 64-row wave group sit in LDS in
// an order that makes a lane's accumulators runs of 8 consecutive features (arm 16: as shipped; arm 32: wperm32 below).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_raw;
typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ bf16_raw f2bf(float f) { return __builtin_bit_cast(bf16_raw, (bf16_t)f); }
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
__device__ __forceinline__ int wperm16(int p) { return (p >> 5) * 32 + ((p & 15) >> 2) * 8 + ((p >> 4) & 1) * 4 + (p & 3); }
__device__ __forceinline__ int lane_feat16(int fq, int t) { return (t >> 1) * 32 + fq * 8 + (t & 1) * 4; }
// arm 32: LDS row ft*32 + rho of a wave group, rho = g*8 + h*4 + q = the MFMA's output row (the lane of half h holds rows
// g*8 + h*4 + q, g = 0..3, q = 0..3), takes W row ft*32 + (g>>1)*16 + h*8 + (g&1)*4 + q: a lane's values g = 0, 1 are 8
// consecutive features at h*8, g = 2, 3 the same + 16
__device__ __forceinline__ int wperm32(int p) {
  const int ft = p >> 5, g = (p >> 3) & 3, h = (p >> 2) & 1, q = p & 3;
  return ft * 32 + (g >> 1) * 16 + h * 8 + (g & 1) * 4 + q;
}
__device__ __forceinline__ uint4 pack8(const float* v) {
  uint4 o;
  o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
  o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
  o.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16);
  o.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
  return o;
}

struct Args {
  const bf16_raw* A;
  const bf16_raw* W;
  bf16_raw* C;
  int M, N, K, band;
};

template <bool M32>
__global__ void __launch_bounds__(512, 2) pp_kernel(Args g) {
  constexpr int TM = 256, TN = 256, BK = 64, NTHR = 512, RS = NTHR / 8, TB = TM * 128;
  constexpr int WBASE = 3 * TB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 2, wn = wid & 3;
  const int tiles_m = g.M / TM, tiles_n = g.N / TN;
  const int nwg = tiles_m * tiles_n;
  const int nkt = g.K / BK;
  const int band = g.band;
  const int xcd = blockIdx.x & 7, lane_in_xcd = blockIdx.x >> 3, per_xcd_wg = gridDim.x >> 3;
  const int q = nwg >> 3, rem = nwg & 7;
  const int xbeg = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
  const int xend = xbeg + (xcd < rem ? q + 1 : q);
  int t_lin = xbeg + lane_in_xcd;
  if (t_lin >= xend) return;
  const int wu = __builtin_amdgcn_readfirstlane(wid);
  const bool late = wu >= 4;
  const int lrow = tid >> 3, lkc = tid & 7;
  const int src_kc = swz(lrow, lkc);
  const int wrow = M32 ? wperm32(lrow) : wperm16(lrow);
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
      AO[k_] = (uint32_t)(((M0 + lrow + k_ * RS) * (long)g.K + src_kc * 8) * 2);           \
      WO[k_] = (uint32_t)(((N0 + k_ * 64 + wrow) * (long)g.K + src_kc * 8) * 2);           \
    }                                                                                      \
  } while (0)
#define PGLDS(base, off, off_lds) \
  __builtin_amdgcn_global_load_lds((gbl_char*)((base) + (off)), (lds_char*)(smem + (off_lds)), 16, 0, 0)
#define PIN() __builtin_amdgcn_sched_barrier(0)
#define BAR() do { PIN(); __builtin_amdgcn_s_barrier(); PIN(); } while (0)

  // fragment read offsets.  arm 16: lane (fr = l & 15, fq = l >> 4) reads row fr (+ 16 t), chunk ks * 4 + fq (two k-steps of
  // 32).  arm 32: lane (r = l & 31, h = l >> 5) reads row r (+ 32 t), chunk ks * 2 + h (four k-steps of 16).
  const int fr = lane & 15, fq = lane >> 4, r32 = lane & 31, h32 = lane >> 5;
  uint32_t a_ro[4], w_ro[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    if (M32) {
      a_ro[ks] = (uint32_t)((wm * 128 + r32) * 128 + (swz(r32, ks * 2 + h32) << 4));
      w_ro[ks] = (uint32_t)(WBASE + (wn * 64 + r32) * 128 + (swz(r32, ks * 2 + h32) << 4));
    } else {
      a_ro[ks] = (uint32_t)((wm * 128 + fr) * 128 + (swz(fr, (ks & 1) * 4 + fq) << 4));
      w_ro[ks] = (uint32_t)(WBASE + (wn * 64 + fr) * 128 + (swz(fr, (ks & 1) * 4 + fq) << 4));
    }
  }
  // (rows r + 16 t / r + 32 t: bits 1..3 of the row are those of r when t * 16 or t * 32 is added -> the swizzle term is the same)
#define LDA16(ks, j) (*reinterpret_cast<const s16x8*>(smem + sa_ + a_ro[ks] + (j) * 2048))
#define LDW16(ks, i) (*reinterpret_cast<const s16x8*>(smem + sw_ + w_ro[ks] + (i) * 2048))
#define LDA32(ks, j) (*reinterpret_cast<const s16x8*>(smem + sa_ + a_ro[ks] + (j) * 4096))
#define LDW32(ks, i) (*reinterpret_cast<const s16x8*>(smem + sw_ + w_ro[ks] + (i) * 4096))

  TILE_COORDS(t_lin, m0, n0);
  TILE_OFFS(m0, n0, ao, wo);
  {
    const char* ab = reinterpret_cast<const char*>(g.A);
    const char* wb = reinterpret_cast<const char*>(g.W);
    const int la = wu * 1024, lw = WBASE + wu * 1024;
#pragma unroll
    for (int k = 0; k < 4; ++k) PGLDS(ab, ao[k], la + k * RS * 128);
#pragma unroll
    for (int k = 0; k < 4; ++k) PGLDS(wb, wo[k], lw + k * RS * 128);
#pragma unroll
    for (int k = 0; k < 4; ++k) PGLDS(ab + BK * 2, ao[k], TB + la + k * RS * 128);
#pragma unroll
    for (int k = 0; k < 4; ++k) PGLDS(wb + BK * 2, wo[k], TB + lw + k * RS * 128);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (late) __builtin_amdgcn_s_barrier();
  int aslot = 0, wslot = 0;
  while (true) {
    const int t_next = t_lin + per_xcd_wg;
    const bool has_next = t_next < xend;
    long nm0 = 0, nn0 = 0;
    uint32_t nao[4] = {ao[0], ao[1], ao[2], ao[3]}, nwo[4] = {wo[0], wo[1], wo[2], wo[3]};
    if (has_next) {
      TILE_COORDS(t_next, nm0, nn0);
      TILE_OFFS(nm0, nn0, nao, nwo);
    }
    uint32_t ca[4] = {ao[0], ao[1], ao[2], ao[3]}, cw[4] = {wo[0], wo[1], wo[2], wo[3]};
    f32x4 acc16[4][8];        // arm 16: [feature tile of 16][token tile of 16]
    f32x16 acc32[2][4];       // arm 32: [feature tile of 32][token tile of 32]
    if (M32) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.f;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int kt = 0; kt < nkt; ++kt) {
      const bool nxt = kt + 2 >= nkt;
      const int k2 = nxt ? kt + 2 - nkt : kt + 2;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ca[k] = nxt ? nao[k] : ca[k];
        cw[k] = nxt ? nwo[k] : cw[k];
      }
      const int a_dst = aslot == 0 ? 2 : aslot - 1;
      const char* ab_ = reinterpret_cast<const char*>(g.A) + (long)k2 * (BK * 2);
      const char* wb_ = reinterpret_cast<const char*>(g.W) + (long)k2 * (BK * 2);
      const int la_ = a_dst * TB + wu * 1024, lw_ = WBASE + wslot * TB + wu * 1024;
      const uint32_t sa_ = aslot * TB, sw_ = wslot * TB;
      if (!M32) {
        s16x8 faA[2][4], faB[2][4], fw[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < 4; ++i) fw[ks][i] = LDW16(ks, i);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int j = 0; j < 4; ++j) faA[ks][j] = LDA16(ks, j);
#pragma unroll
        for (int k = 0; k < 4; ++k) PGLDS(ab_, ca[k], la_ + k * RS * 128);
        BAR();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
              acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ks][i], faA[ks][j], acc16[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        BAR();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int j = 0; j < 4; ++j) faB[ks][j] = LDA16(ks, 4 + j);
#pragma unroll
        for (int k = 0; k < 4; ++k) PGLDS(wb_, cw[k], lw_ + k * RS * 128);
        PIN();
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        BAR();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
              acc16[i][4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ks][i], faB[ks][j], acc16[i][4 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        BAR();
      } else {
        s16x8 faA[4][2], faB[4][2], fw[4][2];      // [k-step of 16][tile of 32]
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int i = 0; i < 2; ++i) fw[ks][i] = LDW32(ks, i);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int j = 0; j < 2; ++j) faA[ks][j] = LDA32(ks, j);
#pragma unroll
        for (int k = 0; k < 4; ++k) PGLDS(ab_, ca[k], la_ + k * RS * 128);
        BAR();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
              acc32[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ks][i], faA[ks][j], acc32[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        BAR();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int j = 0; j < 2; ++j) faB[ks][j] = LDA32(ks, 2 + j);
#pragma unroll
        for (int k = 0; k < 4; ++k) PGLDS(wb_, cw[k], lw_ + k * RS * 128);
        PIN();
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        BAR();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
              acc32[i][2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[ks][i], faB[ks][j], acc32[i][2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        BAR();
      }
      aslot = aslot == 2 ? 0 : aslot + 1;
      wslot ^= 1;
    }
    // ---- epilogue: plain bf16 stores, 16 bytes per access, straight from registers
    {
      bf16_raw* Cb = g.C + (m0 + wm * 128) * (long)g.N + n0 + wn * 64;
      if (M32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bf16_raw* row = Cb + (long)(j * 32 + r32) * g.N;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            float v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = acc32[i][j][e];
            // e = g * 4 + q: g = 0, 1 -> features i*32 + h*8 + (g&1)*4 + q; g = 2, 3 -> + 16
            *reinterpret_cast<uint4*>(row + i * 32 + h32 * 8) = pack8(v);
            *reinterpret_cast<uint4*>(row + i * 32 + 16 + h32 * 8) = pack8(v + 8);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          bf16_raw* row = Cb + (long)(j * 16 + fr) * g.N;
          float v[16];
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[t * 4 + e] = acc16[t][j][e];
          *reinterpret_cast<uint4*>(row + lane_feat16(fq, 0)) = pack8(v);
          *reinterpret_cast<uint4*>(row + lane_feat16(fq, 2)) = pack8(v + 8);
        }
      }
    }
    if (!has_next) break;
    t_lin = t_next;
    m0 = nm0; n0 = nn0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { ao[k] = nao[k]; wo[k] = nwo[k]; }
  }
  if (!late) __builtin_amdgcn_s_barrier();
}

extern "C" int pp_gemm(int arm, const uint16_t* A, const uint16_t* W, uint16_t* C, int M, int N, int K, void* stream) {
  if (M % 256 || N % 256 || K % 64 || K < 128) return -1;
  if ((long)M * K * 2 >= (1L << 32) || (long)N * K * 2 >= (1L << 32)) return -2;
  const int tiles_m = M / 256, tiles_n = N / 256;
  if (tiles_m * tiles_n < 8) return -3;
  Args g{A, W, C, M, N, K, tiles_n <= 16 ? 1 : (tiles_m <= 16 ? tiles_m : 4)};
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)pp_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    (void)hipFuncSetAttribute((const void*)pp_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    attr = true;
  }
  int grid = 256;
  if (tiles_m * tiles_n < grid) grid = (tiles_m * tiles_n + 7) / 8 * 8;
  if (arm == 32) pp_kernel<true><<<grid, 512, 163840, (hipStream_t)stream>>>(g);
  else pp_kernel<false><<<grid, 512, 163840, (hipStream_t)stream>>>(g);
  return hipGetLastError() == hipSuccess ? 0 : -4;
}
