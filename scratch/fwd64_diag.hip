// Diagnostic harness (scratch only, never the shipped library): attn_fwd64's generated body with s_memtime stamps at
// entry / after the prologue / after the first iteration / after the loop / after the last iteration / at the end, one
// 64-byte record per wave.  Built and driven by scratch/fwd64_stamps.py.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "attn_fwd64_diag_body.inc"
#ifdef DIAG_ACC   // generator run with --acc: the macros carry the Q-variant's names
#define ATTN_FWD64_BODY ATTN_FWD64Q_BODY
#define ATTN_FWD64_CLOBBERS ATTN_FWD64Q_CLOBBERS
#endif
extern "C" __global__ void __launch_bounds__(256, 1)
fwd64_diag_kernel(const uint16_t* Q, const uint16_t* K, const uint16_t* Vt, uint16_t* O, unsigned long long* dbg, int B, int H,
                  int S, long ldo, long o_bstride, float scale_log2e) {
  const int nq = S >> 8;
  const int nwg = nq * H * B;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, rr = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
  }
  const int qt = bid % nq, bh = bid / nq, b = bh / H, hh = bh - b * H;
  const unsigned long long qp = (unsigned long long)(Q + ((long)bh * S + qt * 256) * 128);
  const unsigned long long kp = (unsigned long long)(K + (long)bh * S * 128);
  const unsigned long long vp = (unsigned long long)(Vt + (long)bh * 128 * S);
  const unsigned long long op = (unsigned long long)(O + (long)b * o_bstride + (long)(qt * 256) * ldo + hh * 128);
  const unsigned long long dp = (unsigned long long)(dbg + (long)blockIdx.x * 32);
  const int ntiles = S >> 6;
  asm volatile(ATTN_FWD64_BODY
               :
               : [tid] "v"(threadIdx.x), [q_lo] "s"((unsigned)qp), [q_hi] "s"((unsigned)(qp >> 32)), [k_lo] "s"((unsigned)kp),
                 [k_hi] "s"((unsigned)(kp >> 32)), [v_lo] "s"((unsigned)vp), [v_hi] "s"((unsigned)(vp >> 32)),
                 [o_lo] "s"((unsigned)op), [o_hi] "s"((unsigned)(op >> 32)), [l_lo] "s"(0u), [l_hi] "s"(0u), [sp2] "s"(S * 2),
                 [ldo2] "s"((int)(ldo * 2)), [cs] "s"(scale_log2e), [nloop] "s"((ntiles - 2) >> 1),
                 [kmax] "s"((ntiles - 1) * 16384), [vmax] "s"((ntiles - 1) * 128), [d_lo] "s"((unsigned)dp),
                 [d_hi] "s"((unsigned)(dp >> 32)), [nblk] "s"(1), [qt0] "s"(qt), [hh0] "s"(hh), [b0] "s"(b), [nq] "s"(nq), [nh] "s"(H),
                 [kstep] "s"(S * 256), [ostep] "s"((int)(ldo * 512)), [obs] "s"((int)(o_bstride * 2)),
                 [ob_lo] "s"((unsigned)(unsigned long long)O), [ob_hi] "s"((unsigned)((unsigned long long)O >> 32)), [sq] "s"(0),
                 [dbh] "s"(0), [qstride] "s"(0), [lstride] "s"(0)
               : ATTN_FWD64_CLOBBERS);
}
extern "C" int fwd64_diag(const uint16_t* Q, const uint16_t* K, const uint16_t* Vt, uint16_t* O, unsigned long long* dbg, int B,
                          int H, int S, long ldo, long o_bstride, float scale, void* stream) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)fwd64_diag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    attr = true;
  }
  fwd64_diag_kernel<<<(S / 256) * H * B, 256, 65536, (hipStream_t)stream>>>(Q, K, Vt, O, dbg, B, H, S, ldo, o_bstride,
                                                                             scale * 1.4426950408889634f);
  return (int)hipGetLastError();
}
