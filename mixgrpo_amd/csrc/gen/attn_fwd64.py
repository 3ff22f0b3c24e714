#!/usr/bin/env python3
"""Generator of the hand-placed gfx950 instruction stream of `attn_fwd64_kernel` (mixgrpo_amd/csrc/attention.hip).

Joint text+image attention forward of the FLUX MMDiT (replaces F.scaled_dot_product_attention at the reference's call
sites fastvideo/utils/sampling_utils.py:68-82 and fastvideo/train_grpo_flux.py:134-144), for S % 256 == 0.

Why a generated stream: the 8-wave x 32-query kernel is held by the chip's clock, not by its schedule (DESIGN.md section 6,
round 2); what lowers the energy per FLOP is a wave that owns 64 queries -- two independent 32-query chains A and B that share
every K / V^T fragment it reads from LDS -- at ONE wave per SIMD with all 512 registers.  hipcc cannot be made to schedule
that (round 1: "register-bound under hipcc"), so the whole body is placed here, one MFMA "gap" at a time:

  workgroup = 4 waves = 256 queries of one (batch, head); wave = 64 queries = chains A (first 32) and B.
  K / V^T tiles of 64 keys arrive by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, swizzled through the
  per-lane SOURCE address) into two slots each (64 KiB of LDS), one s_barrier per tile.
  Registers: O_A a[0:63], O_B a[64:127], Q_A a[128:159], Q_B a[160:191], the tile's 16 K fragments a[192:255];
             S_A v[4:35], S_B v[36:67], P_A v[68:83], P_B v[84:99], the tile's 16 V^T fragments v[100:163].
  Iteration i (one tile, 64 MFMAs) = two segments of 32 MFMA gaps:
     segment 1: MFMA  P V of chain B, tile i-1 (16)  +  S^T = K Q^T of chain B, tile i (16)
                VALU  softmax of chain A, tile i          LDS  V^T(i) fragments (after their last use)   DMA  K(i+2), V^T(i+1)
     segment 2: MFMA  P V of chain A, tile i (16)    +  S^T of chain A, tile i+1 (16)
                VALU  softmax of chain B, tile i          LDS  K(i+1) fragments
  i.e. one chain's softmax is issued while the other chain's MFMAs execute, and every fragment is read from LDS once per 64
  queries.  Per gap: 1 MFMA, 1 v_exp_f32, <= 3-4 other VALU / LDS / DMA instructions (MI355X_MICROARCH.md: an MFMA gap hides
  <= 5 single-issue instructions, at most one of them transcendental).

Softmax: p = exp2(s * c - m * c), c = scale * log2(e).  m is the row maximum of the FIRST tile; afterwards no maximum is
formed at all: a tile's row sums are checked against 2^40 (one v_cmp per chain and tile) and only if some row exceeds it --
a later key outscores the first tile's maximum by > 27 nats -- the (out-of-line) fix-up recomputes that tile's maximum,
rescales O and l and redoes the tile's exponentials.  P <= 2^40 is exact in bf16 / fp32 relative precision; the fix-up has
its own forced test (tests/test_hip_mmdit.py::test_attention_fwd64_rescale_path).

The stream is checked on the CPU before it ever runs: tests/asm_emu.py interprets it (4 waves, LDS, waitcnt-visible data)
against the oracle and a static pass checks the gfx950 software hazards (MFMA result -> VALU read 12 wait states, ...).

Second variant (round 4, `ACC`, attn_fwd64q_body.inc, kernel attn_fwd64q_kernel): for a Q that the producer has ALREADY scaled by
c (mgx_qk_norm_rope_fwd's q_scale: one bf16 rounding of q c instead of one of q), with m SUBTRACTED BY THE MATRIX PIPE: from
the second tile on, the first of a score tile's eight MFMAs takes C = a block of 16 registers holding -m (one block per chain,
written once when the first tile's maximum is known, moved by the fix-up), so the accumulator comes out as s c - m c and the
softmax's `v_fma_f32 t, s, c, -mc` -- one of the ~3.5 single-issue instructions per MFMA gap in a loop that is issue-bound
at 2552 cycles per iteration against the pipe's 2048 -- disappears: v_exp_f32 reads the accumulator directly.  The first tile
(scores needed before m exists) still subtracts, with v_sub_f32.

Run `python mixgrpo_amd/csrc/gen/attn_fwd64.py` to rewrite mixgrpo_amd/csrc/attn_fwd64_body.inc and attn_fwd64q_body.inc
(build.py does).
"""
import os
import sys

# ------------------------------------------------------------------------------------------------ register map
O_A, O_B = 0, 64          # a: 4 d-tiles x 16
Q_A, Q_B = 128, 160       # a: 8 k-steps x 4
KF = 192                  # a: 16 fragments (n = 8 kb + ks) x 4
S_A, S_B = 4, 36          # v: 2 key blocks x 16
P_A, P_B = 68, 84         # v: 4 k-steps x 4 (bf16 pairs)
VF = 100                  # v: 16 fragments (n = 4 s + dt) x 4
KA = 164                  # v: 8 K fragment read addresses (per k-step)
VA = 172                  # v: 4 V^T fragment read addresses (per key step)
KS = 176                  # v: 4 K DMA source offsets
VS = 180                  # v: 4 V^T DMA source offsets
T = 184                   # v: 8 rotating temporaries
M_A, M_B = 192, 193       # running max (raw scores)
MC_A, MC_B = 194, 195     # m * c
L_A, L_B = 196, 197       # row sums (this lane half's keys)
PS0, PS1 = 198, 199       # tile row sums
X = 200                   # v[200:231]: prologue / epilogue / fix-up scratch
QOFF_A, QOFF_B = 232, 233
OOFF_A, OOFF_B = 234, 235
LOFF = 236
V_LAST = 239
NM_A, NM_B = 240, X + 16  # ACC variant: v[240:255] / v[216:231] = 16 x (-m) of chain A / B, the SrcC of a score tile's first MFMA
                          # (X + 16 .. X + 31 is otherwise only the epilogue's staging: dead while the blocks are live)
ACC = False               # set by generate(acc=True)
DBG_OFF, DBG_LO = 237, 238   # v237, v[238:239]: diagnostic builds only

# scalars (copied from the asm operands into fixed registers, all clobbered)
sW, sWOFF = 64, 65
sKP, sVP = 66, 68         # pairs: DMA base pointers of the tiles being fetched
sLOOP = 70
sKT, sVT = 71, 72         # byte offsets of the next K / V^T tiles to fetch
sTMP = 73
sQ, sK, sV, sO, sL = 74, 76, 78, 80, 82
sSP2, sLDO2, sCS, sNLOOP, sKMAX, sVMAX = 84, 85, 86, 87, 88, 89
sRET = 90                 # pair: diagnostic stamps; the K(1) prefetch pointer of the next block
sDBG = 92                 # pair (diagnostic builds)
sLEFT, sQT, sHH, sADV, sWRAP = 94, 95, 96, 97, 98       # block loop: blocks left, (q-tile, head) of the block, flags
sQTN, sBB = 84, 85               # next q-tile, batch (sSP2 / sLDO2's registers: those are read by the address set-up only)
S_FIRST, S_LAST = 64, 99

K_BASE, V_BASE = 0, 32768
SLOT = 16384
BIG = "0x53800000"        # 2^40


def v(i):
    return f"v{i}"


def vr(i, n):
    return f"v[{i}:{i + n - 1}]"


def a(i):
    return f"a{i}"


def ar(i, n):
    return f"a[{i}:{i + n - 1}]"


def s(i):
    return f"s{i}"


def sr(i, n):
    return f"s[{i}:{i + n - 1}]"


class Chain:
    def __init__(self, name, O, Q, S, P, M, MC, L, NM):
        self.name, self.O, self.Q, self.S, self.P, self.M, self.MC, self.L, self.NM = name, O, Q, S, P, M, MC, L, NM


CA = Chain("A", O_A, Q_A, S_A, P_A, M_A, MC_A, L_A, NM_A)
CB = Chain("B", O_B, Q_B, S_B, P_B, M_B, MC_B, L_B, NM_B)


class Asm:
    def __init__(self):
        self.lines = []
        self.nlabel = 0

    def e(self, text):
        self.lines.append("  " + text)

    def c(self, text):
        self.lines.append("  ; " + text)

    def label(self, name):
        self.lines.append(f"{name}:")

    def new_label(self, stem):
        self.nlabel += 1
        return f".L{stem}_{self.nlabel}_%="

    def text(self):
        return "\n".join(self.lines) + "\n"


MFMA = "v_mfma_f32_32x32x16_bf16"


# ------------------------------------------------------------------------------------------------ building blocks
def mfma_pv(ch, n):
    """n = 4 s + dt: O[dt] += Vt fragment (s, dt) x P[s]."""
    sidx, dt = n >> 2, n & 3
    o = ar(ch.O + 16 * dt, 16)
    return f"{MFMA} {o}, {vr(VF + 4 * n, 4)}, {vr(ch.P + 4 * sidx, 4)}, {o}"


def mfma_qk(ch, n, first_tile=False):
    """n = 8 kb + ks: S[kb] (+)= K fragment (kb, ks) x Q[ks].  ACC: a tile's first product starts from -m (the chain's NM
    block) instead of 0 -- except in the block's first tile, whose scores define m."""
    kb, ks = n >> 3, n & 7
    sreg = vr(ch.S + 16 * kb, 16)
    c0 = vr(ch.NM, 16) if (ACC and not first_tile) else "0"
    return f"{MFMA} {sreg}, {ar(KF + 4 * n, 4)}, {ar(ch.Q + 4 * ks, 4)}, {c0 if ks == 0 else sreg}"


def ds_read_k(n, slot):
    kb, ks = n >> 3, n & 7
    return f"ds_read_b128 {ar(KF + 4 * n, 4)}, {v(KA + ks)} offset:{K_BASE + slot * SLOT + kb * 8192}"


def ds_read_v(n, slot):
    sidx, dt = n >> 2, n & 3
    return f"ds_read_b128 {vr(VF + 4 * n, 4)}, {v(VA + sidx)} offset:{slot * SLOT + dt * 4096}"   # V_BASE is in the address


def dma_piece(kind, p, slot, ptr=None):
    """(m0 write, load) of piece p (0..3) of this wave's quarter of a K ('k') or V^T ('v') tile."""
    base = (K_BASE if kind == "k" else V_BASE) + slot * SLOT + p * 1024
    src = (KS if kind == "k" else VS) + p
    ptr = ptr if ptr is not None else (sKP if kind == "k" else sVP)
    return (f"s_add_i32 m0, {s(sWOFF)}, {base}", f"global_load_lds_dwordx4 {v(src)}, {sr(ptr, 2)}")


def dma_setup():
    """Base pointers of the next K / V^T tiles (clamped to the last tile: the one over-fetch at the end re-reads it)."""
    return [f"s_add_u32 {s(sKP)}, {s(sK)}, {s(sKT)}", f"s_addc_u32 {s(sKP + 1)}, {s(sK + 1)}, 0",
            f"s_add_u32 {s(sVP)}, {s(sV)}, {s(sVT)}", f"s_addc_u32 {s(sVP + 1)}, {s(sV + 1)}, 0"]


def dma_advance():
    return [f"s_add_u32 {s(sKT)}, {s(sKT)}, {SLOT}", f"s_min_u32 {s(sKT)}, {s(sKT)}, {s(sKMAX)}",
            f"s_add_u32 {s(sVT)}, {s(sVT)}, 128", f"s_min_u32 {s(sVT)}, {s(sVT)}, {s(sVMAX)}"]


def block_advance_loads():
    """Top of a block's last iteration: Q / K / V^T pointers of the workgroup's NEXT block -- `stride` blocks further in the
    (batch, head, q-tile) order, so that the workgroups of an XCD work on neighbouring q-tiles of the same heads at the same
    time (K / V shared in L2); the workgroup's last block stays where it is -- and the pointers of its first three tiles."""
    return [f"s_cmp_gt_u32 {s(sLEFT)}, 1", f"s_cselect_b32 {s(sADV)}, 1, 0",
            f"s_mul_i32 {s(sTMP)}, {s(sADV)}, %[qstride]",                        # stride q-tiles of Q (256 rows x 256 bytes each)
            f"s_add_u32 {s(sQ)}, {s(sQ)}, {s(sTMP)}", f"s_addc_u32 {s(sQ + 1)}, {s(sQ + 1)}, 0",
            f"s_mul_i32 {s(sTMP)}, {s(sADV)}, %[sq]", f"s_add_u32 {s(sQTN)}, {s(sQT)}, {s(sTMP)}",    # q-tile + stride % nq
            f"s_mul_i32 {s(sWRAP)}, {s(sADV)}, %[dbh]",                           # heads advanced: stride / nq ...
            f"s_cmp_ge_u32 {s(sQTN)}, %[nq]", f"s_cselect_b32 {s(sTMP)}, %[nq], 0",
            f"s_sub_u32 {s(sQTN)}, {s(sQTN)}, {s(sTMP)}", f"s_cmp_lg_u32 {s(sTMP)}, 0",
            f"s_addc_u32 {s(sWRAP)}, {s(sWRAP)}, 0",                              # ... + 1 when the q-tile index wrapped
            f"s_mul_i32 {s(sTMP)}, {s(sWRAP)}, %[kstep]",                         # per head: S rows of K, 128 rows of V^T
            f"s_add_u32 {s(sK)}, {s(sK)}, {s(sTMP)}", f"s_addc_u32 {s(sK + 1)}, {s(sK + 1)}, 0",
            f"s_add_u32 {s(sV)}, {s(sV)}, {s(sTMP)}", f"s_addc_u32 {s(sV + 1)}, {s(sV + 1)}, 0",
            f"s_mov_b32 {s(sKP)}, {s(sK)}", f"s_mov_b32 {s(sKP + 1)}, {s(sK + 1)}",
            f"s_mov_b32 {s(sVP)}, {s(sV)}", f"s_mov_b32 {s(sVP + 1)}, {s(sV + 1)}",
            f"s_add_u32 {s(sRET)}, {s(sK)}, {SLOT}", f"s_addc_u32 {s(sRET + 1)}, {s(sK + 1)}, 0",
            f"s_mov_b32 {s(sKT)}, {2 * SLOT}", f"s_mov_b32 {s(sVT)}, 128"]        # the next block's iteration 0 fetches K(2), V^T(1)


def block_advance_stores(A):
    """Behind the epilogue: (batch, head, q-tile) of the next block, its O / lse pointers; one block less to go."""
    A.e(f"s_mov_b32 {s(sQT)}, {s(sQTN)}")
    A.e(f"s_add_u32 {s(sHH)}, {s(sHH)}, {s(sWRAP)}")
    A.e(f"s_cmp_ge_u32 {s(sHH)}, %[nh]")
    A.e(f"s_cselect_b32 {s(sTMP)}, %[nh], 0")
    A.e(f"s_sub_u32 {s(sHH)}, {s(sHH)}, {s(sTMP)}")
    A.e(f"s_cmp_lg_u32 {s(sTMP)}, 0")
    A.e(f"s_addc_u32 {s(sBB)}, {s(sBB)}, 0")
    # O = base + batch * (bytes per batch) + q-tile * (bytes of 256 rows) + head * 256
    A.e(f"s_mul_i32 {s(sO)}, {s(sBB)}, %[obs]")
    A.e(f"s_mul_hi_u32 {s(sO + 1)}, {s(sBB)}, %[obs]")
    A.e(f"s_mul_i32 {s(sTMP)}, {s(sQT)}, %[ostep]")
    A.e(f"s_add_u32 {s(sO)}, {s(sO)}, {s(sTMP)}")
    A.e(f"s_addc_u32 {s(sO + 1)}, {s(sO + 1)}, 0")
    A.e(f"s_lshl_b32 {s(sTMP)}, {s(sHH)}, 8")
    A.e(f"s_add_u32 {s(sO)}, {s(sO)}, {s(sTMP)}")
    A.e(f"s_addc_u32 {s(sO + 1)}, {s(sO + 1)}, 0")
    A.e(f"s_add_u32 {s(sO)}, {s(sO)}, %[ob_lo]")
    A.e(f"s_addc_u32 {s(sO + 1)}, {s(sO + 1)}, %[ob_hi]")
    nol = A.new_label("nolnext")
    A.e(f"s_cmp_eq_u64 {sr(sL, 2)}, 0")
    A.e(f"s_cbranch_scc1 {nol}")
    A.e(f"s_mul_i32 {s(sTMP)}, {s(sADV)}, %[lstride]")
    A.e(f"s_add_u32 {s(sL)}, {s(sL)}, {s(sTMP)}")
    A.e(f"s_addc_u32 {s(sL + 1)}, {s(sL + 1)}, 0")
    A.label(nol)
    A.e(f"s_sub_u32 {s(sLEFT)}, {s(sLEFT)}, 1")


def softmax_gaps(ch, ngaps=32, first_tile=False):
    """VALU stream of one chain's tile softmax, as `ngaps` lists (one per MFMA gap) + a tail list.

    Element e (0..31) = S register e; fma(e) two gaps before exp(e), the sum one gap after, cvt_pk(k) one gap after the
    exponential of its odd element; temporaries rotate through T[e % 8].  ACC: the accumulator already holds s c - m c (no
    fma at all; the exponential reads it directly), except in the block's first tile: v_sub_f32 with m."""
    def t(e):
        return v(T + (e & 7))

    def fma(e):
        if ACC:
            return f"v_sub_f32 {t(e)}, {v(ch.S + e)}, {v(ch.MC)}" if first_tile else None
        return f"v_fma_f32 {t(e)}, {v(ch.S + e)}, {s(sCS)}, -{v(ch.MC)}"

    def exp(e):
        if ACC and not first_tile:
            return f"v_exp_f32 {t(e)}, {v(ch.S + e)}"
        return f"v_exp_f32 {t(e)}, {t(e)}"

    def add(e):
        k = e >> 1
        ps = v(PS0 if (k & 1) == 0 else PS1)
        if k < 2:                          # the first pair of each accumulator initialises it
            return f"v_add_f32 {ps}, {t(e - 1)}, {t(e)}" if (e & 1) else None
        return f"v_add_f32 {ps}, {ps}, {t(e)}"

    def cvt(k):
        return f"v_cvt_pk_bf16_f32 {v(ch.P + k)}, {t(2 * k)}, {t(2 * k + 1)}"

    gaps = [[] for _ in range(32)]
    gaps[0] += [x for x in (fma(0), fma(1), fma(2), fma(3)) if x] + [exp(0), exp(1)]
    for g in range(1, 32):
        if g + 1 < 32:
            gaps[g].append(exp(g + 1))
        if g + 3 < 32 and fma(g + 3):
            gaps[g].append(fma(g + 3))
        if g == 1:
            gaps[g].append(add(1))          # e = 0 has no instruction of its own (pair sum)
        else:
            x = add(g)
            if x:
                gaps[g].append(x)
        if g & 1:
            gaps[g].append(cvt((g - 1) >> 1))
    if ngaps == 16:                         # first iteration's short segment: two softmax gaps per MFMA gap
        gaps = [gaps[2 * i] + gaps[2 * i + 1] for i in range(16)]
    tail = [f"v_add_f32 {v(PS0)}, {v(PS0)}, {v(PS1)}"]
    return gaps, tail


def max_prefix(ch):
    """First tile: m = row maximum of the tile (both lane halves), mc = m * c."""
    t0, t1 = v(X), v(X + 1)
    out = [f"v_max_f32 {t0}, {v(ch.S)}, {v(ch.S + 1)}"]
    for e in range(2, 32, 2):
        out.append(f"v_max3_f32 {t0}, {t0}, {v(ch.S + e)}, {v(ch.S + e + 1)}")
    out += [f"v_mov_b32 {t1}, {t0}", "s_nop 1", f"v_permlane32_swap_b32 {t0}, {t1}",
            f"v_max_f32 {v(ch.M)}, {t0}, {t1}",
            f"v_mov_b32 {v(ch.MC)}, {v(ch.M)}" if ACC else f"v_mul_f32 {v(ch.MC)}, {v(ch.M)}, {s(sCS)}"]
    if ACC:                                  # (scores are exponents already: mc == m) the SrcC block of every later score tile
        out += [f"v_sub_f32 {v(ch.NM + j)}, 0, {v(ch.M)}" for j in range(16)]
    return out


def fixup(A, ch, back):
    """Out-of-line: some row sum of the tile exceeded 2^40.  New maximum from the (intact) S registers, O and l rescaled
    exactly once, the tile's P and row sums redone.  Not scheduled: it practically never runs."""
    t0, t1, al = v(X), v(X + 1), v(X + 2)
    A.e("s_nop 7")
    A.e("s_nop 7")                                          # every MFMA that wrote O / S of this chain has retired
    if ACC:
        return fixup_acc(A, ch, back)
    A.e(f"v_max_f32 {t0}, {v(ch.S)}, {v(ch.S + 1)}")
    for e in range(2, 32, 2):
        A.e(f"v_max3_f32 {t0}, {t0}, {v(ch.S + e)}, {v(ch.S + e + 1)}")
    A.e(f"v_mov_b32 {t1}, {t0}")
    A.e("s_nop 1")
    A.e(f"v_permlane32_swap_b32 {t0}, {t1}")
    A.e(f"v_max3_f32 {t0}, {t0}, {t1}, {v(ch.M)}")          # m_new
    A.e(f"v_sub_f32 {t1}, {v(ch.M)}, {t0}")
    A.e(f"v_mul_f32 {t1}, {t1}, {s(sCS)}")
    A.e(f"v_exp_f32 {al}, {t1}")                            # alpha = 2^((m - m_new) c)
    A.e(f"v_mov_b32 {v(ch.M)}, {t0}")
    A.e(f"v_mul_f32 {v(ch.MC)}, {t0}, {s(sCS)}")
    A.e(f"v_mul_f32 {v(ch.L)}, {v(ch.L)}, {al}")
    for blk in range(0, 64, 8):
        for j in range(8):
            A.e(f"v_accvgpr_read_b32 {v(X + 8 + j)}, {a(ch.O + blk + j)}")
        for j in range(8):
            A.e(f"v_mul_f32 {v(X + 8 + j)}, {v(X + 8 + j)}, {al}")
        for j in range(8):
            A.e(f"v_accvgpr_write_b32 {a(ch.O + blk + j)}, {v(X + 8 + j)}")
    for k in range(16):                                     # P and row sums again, at the new maximum
        ta, tb = v(X + 8), v(X + 9)
        A.e(f"v_fma_f32 {ta}, {v(ch.S + 2 * k)}, {s(sCS)}, -{v(ch.MC)}")
        A.e(f"v_fma_f32 {tb}, {v(ch.S + 2 * k + 1)}, {s(sCS)}, -{v(ch.MC)}")
        A.e(f"v_exp_f32 {ta}, {ta}")
        A.e(f"v_exp_f32 {tb}, {tb}")
        A.e("s_nop 0")
        if k == 0:
            A.e(f"v_add_f32 {v(PS0)}, {ta}, {tb}")
        else:
            A.e(f"v_add_f32 {v(PS0)}, {v(PS0)}, {ta}")
            A.e(f"v_add_f32 {v(PS0)}, {v(PS0)}, {tb}")
        A.e(f"v_cvt_pk_bf16_f32 {v(ch.P + k)}, {ta}, {tb}")
    A.e("s_nop 7")                                          # accvgpr / P writes -> the next MFMAs
    A.e(f"s_branch {back}")


def fixup_acc(A, ch, back):
    """ACC variant: the S registers hold s - m already, so r = max(0, row maximum of S) is how far m moves: alpha = 2^-r,
    P = 2^(S - r), m += r, the -m block -= r (the chain's next score tile is issued after this returns)."""
    t0, t1, al = v(X), v(X + 1), v(X + 2)
    A.e(f"v_max_f32 {t0}, {v(ch.S)}, {v(ch.S + 1)}")
    for e in range(2, 32, 2):
        A.e(f"v_max3_f32 {t0}, {t0}, {v(ch.S + e)}, {v(ch.S + e + 1)}")
    A.e(f"v_mov_b32 {t1}, {t0}")
    A.e("s_nop 1")
    A.e(f"v_permlane32_swap_b32 {t0}, {t1}")
    A.e(f"v_max3_f32 {t0}, {t0}, {t1}, 0")                  # r
    A.e(f"v_sub_f32 {t1}, 0, {t0}")
    A.e(f"v_exp_f32 {al}, {t1}")                            # alpha = 2^-r
    A.e(f"v_add_f32 {v(ch.M)}, {v(ch.M)}, {t0}")
    A.e(f"v_mov_b32 {v(ch.MC)}, {v(ch.M)}")
    for j in range(16):
        A.e(f"v_sub_f32 {v(ch.NM + j)}, 0, {v(ch.M)}")
    A.e(f"v_mul_f32 {v(ch.L)}, {v(ch.L)}, {al}")
    for blk in range(0, 64, 8):
        for j in range(8):
            A.e(f"v_accvgpr_read_b32 {v(X + 8 + j)}, {a(ch.O + blk + j)}")
        for j in range(8):
            A.e(f"v_mul_f32 {v(X + 8 + j)}, {v(X + 8 + j)}, {al}")
        for j in range(8):
            A.e(f"v_accvgpr_write_b32 {a(ch.O + blk + j)}, {v(X + 8 + j)}")
    for k in range(16):                                     # P and row sums again, at the new maximum
        ta, tb = v(X + 8), v(X + 9)
        A.e(f"v_sub_f32 {ta}, {v(ch.S + 2 * k)}, {t0}")
        A.e(f"v_sub_f32 {tb}, {v(ch.S + 2 * k + 1)}, {t0}")
        A.e(f"v_exp_f32 {ta}, {ta}")
        A.e(f"v_exp_f32 {tb}, {tb}")
        A.e("s_nop 0")
        if k == 0:
            A.e(f"v_add_f32 {v(PS0)}, {ta}, {tb}")
        else:
            A.e(f"v_add_f32 {v(PS0)}, {v(PS0)}, {ta}")
            A.e(f"v_add_f32 {v(PS0)}, {v(PS0)}, {tb}")
        A.e(f"v_cvt_pk_bf16_f32 {v(ch.P + k)}, {ta}, {tb}")
    A.e("s_nop 7")                                          # accvgpr / P / -m writes -> the next MFMAs
    A.e(f"s_branch {back}")


TIMING_ONLY = set()        # diagnostic variants (wrong results; scratch harness only): "nodma", "nolds", "novalu", "nobarrier"


def segment(A, mfmas, valu_gaps, valu_tail, lds=None, dma=None, pre=None, waits=None, extra=None):
    """Emit one segment: per gap [wait] MFMA, the gap's VALU slice, at most one LDS read, at most one DMA piece.

    lds: {gap: instr}; dma: {gap: (m0 write, load)}; waits: {gap: 's_waitcnt ...'} placed in front of the gap's MFMA."""
    lds, dma, waits, extra = lds or {}, dma or {}, waits or {}, extra or {}
    if "nodma" in TIMING_ONLY:
        dma = {}
    if "nolds" in TIMING_ONLY:
        lds = {}
    if "novalu" in TIMING_ONLY:
        valu_gaps = [[] for _ in valu_gaps]
    for x in pre or []:
        A.e(x)
    for g, m in enumerate(mfmas):
        if g in waits:
            A.e(waits[g])
        A.e(m)
        fill = list(valu_gaps[g]) if g < len(valu_gaps) else []
        if g in dma:
            m0w, ld = dma[g]
            A.e(m0w)
            if fill:
                A.e(fill.pop(0))           # one VALU between the m0 write and the DMA (the required wait state)
            else:
                A.e("s_nop 0")
            A.e(ld)
        for x in fill:
            A.e(x)
        if g in lds:
            A.e(lds[g])
        for x in extra.get(g, []):
            A.e(x)
    for x in valu_tail:
        A.e(x)


def iteration(A, par, fixups, first=False, last=False):
    """One K/V tile i with i & 1 == par.  Slots: K(j), V(j) live in slot j & 1."""
    A.c(f"================ iteration parity {par}{' FIRST' if first else ''}{' LAST' if last else ''}")
    # ---------------- segment 1: MFMA chain B (P V of tile i-1, S^T of tile i), softmax chain A
    mf = ([] if first else [mfma_pv(CB, n) for n in range(16)]) + [mfma_qk(CB, n, first_tile=first) for n in range(16)]
    ng = len(mf)
    vg, vt = softmax_gaps(CA, ng, first_tile=first)
    pre = []
    if first:
        pre = max_prefix(CA)
    dma, lds, extra1, extra2 = {}, {}, {}, {}
    if last:
        # the NEXT block's first tiles and Q fragments (the same block again when this is the workgroup's last: unused):
        # K(0), V^T(0) -> slot 0 and K(1) -> slot 1 are free since the barrier behind iteration nt-2; Q_A since then too, Q_B
        # after its last S^T product in this segment
        pre = block_advance_loads() + pre
        pieces = ([dma_piece("k", p, 0) for p in range(4)] + [dma_piece("v", p, 0) for p in range(4)] +
                  [dma_piece("k", p, 1, ptr=sRET) for p in range(4)])
        for j, pc in enumerate(pieces):
            dma[j] = pc
        for ks in range(8):
            extra1[12 + ks] = [f"global_load_dwordx4 {ar(CA.Q + 4 * ks, 4)}, {v(QOFF_A)}, {sr(sQ, 2)} offset:{32 * ks}"]
            extra2[ks] = [f"global_load_dwordx4 {ar(CB.Q + 4 * ks, 4)}, {v(QOFF_B)}, {sr(sQ, 2)} offset:{32 * ks}"]
    if not last:
        pre = dma_setup() + pre
        pieces = [dma_piece("k", p, par) for p in range(4)] + [dma_piece("v", p, 1 - par) for p in range(4)]
        for j, pc in enumerate(pieces):
            dma[j] = pc
        vt = vt + dma_advance()
    # V^T(i) fragments into the V registers, each after its last use by P V (B, i-1) at gap n
    for n in range(16):
        lds[(n if first else 8 + n)] = ds_read_v(n, par)
    A.c("---- segment 1")
    segment(A, mf, vg, vt, lds=lds, dma=dma, pre=pre, extra=extra1)
    end_of_softmax(A, CA, fixups, first)
    # ---------------- segment 2: MFMA chain A (P V of tile i, S^T of tile i+1), softmax chain B
    mf = [mfma_pv(CA, n) for n in range(16)] + ([] if last else [mfma_qk(CA, n) for n in range(16)])
    vg, vt = softmax_gaps(CB, 32 if not last else 16, first_tile=first)
    pre = max_prefix(CB) if first else []
    lds, waits = {}, {0: "s_waitcnt lgkmcnt(0)"}            # every V^T fragment of the tile has landed
    if not last:
        for n in range(16):
            lds[n] = ds_read_k(n, 1 - par)                  # K(i+1), used from gap 16 on in the order read
        waits[16] = "s_waitcnt lgkmcnt(8)"
        waits[24] = "s_waitcnt lgkmcnt(0)"
    A.c("---- segment 2")
    segment(A, mf, vg, vt, lds=lds, pre=pre, waits=waits, extra=extra2)
    end_of_softmax(A, CB, fixups, first)
    if not last:
        A.e("s_waitcnt vmcnt(0)")                           # this iteration's K(i+2), V^T(i+1) pieces
        if "nobarrier" not in TIMING_ONLY:
            A.e("s_barrier")


def end_of_softmax(A, ch, fixups, first):
    if first:
        A.e(f"v_mov_b32 {v(ch.L)}, {v(PS0)}")
        return
    fix, back = A.new_label(f"fix{ch.name}"), A.new_label(f"back{ch.name}")
    A.e(f"v_cmp_lt_f32 vcc, {BIG}, {v(PS0)}")
    A.e(f"s_cbranch_vccnz {fix}")
    A.label(back)
    A.e(f"v_add_f32 {v(ch.L)}, {v(ch.L)}, {v(PS0)}")
    fixups.append((fix, back, ch))


# ------------------------------------------------------------------------------------------------ prologue / epilogue
def prologue(A):
    A.c("inputs -> fixed scalar registers")
    for dst, name in ((sQ, "q_lo"), (sQ + 1, "q_hi"), (sK, "k_lo"), (sK + 1, "k_hi"), (sV, "v_lo"), (sV + 1, "v_hi"),
                      (sO, "o_lo"), (sO + 1, "o_hi"), (sL, "l_lo"), (sL + 1, "l_hi"), (sSP2, "sp2"), (sLDO2, "ldo2"),
                      (sCS, "cs"), (sNLOOP, "nloop"), (sKMAX, "kmax"), (sVMAX, "vmax")):
        A.e(f"s_mov_b32 {s(dst)}, %[{name}]")
    lane, w, r, h = v(X), v(X + 1), v(X + 2), v(X + 3)
    t0, t1, t2 = v(X + 4), v(X + 5), v(X + 6)
    A.e(f"v_and_b32 {lane}, 63, %[tid]")
    A.e(f"v_lshrrev_b32 {w}, 6, %[tid]")
    A.e(f"v_and_b32 {r}, 31, {lane}")
    A.e(f"v_lshrrev_b32 {h}, 5, {lane}")
    A.e(f"v_readfirstlane_b32 {s(sW)}, {w}")
    A.e(f"s_lshl_b32 {s(sWOFF)}, {s(sW)}, 12")
    A.c("K fragment read addresses: MFMA row r reads tile row pi(r) (bits 2 and 3 of r exchanged), so that the S^T")
    A.c("accumulator's rows are keys 16 s + 8 h + j, the plain V^T chunk order")
    pi, xk, pi8 = v(X + 7), v(X + 8), v(X + 9)
    A.e(f"v_and_b32 {t0}, 0x13, {r}")
    A.e(f"v_and_b32 {t1}, 4, {r}")
    A.e(f"v_lshlrev_b32 {t1}, 1, {t1}")
    A.e(f"v_and_b32 {t2}, 8, {r}")
    A.e(f"v_lshrrev_b32 {t2}, 1, {t2}")
    A.e(f"v_or3_b32 {pi}, {t0}, {t1}, {t2}")
    A.e(f"v_and_b32 {t0}, 15, {pi}")
    A.e(f"v_xor_b32 {xk}, {h}, {t0}")
    A.e(f"v_lshlrev_b32 {pi8}, 8, {pi}")
    for ks in range(8):
        A.e(f"v_xor_b32 {t0}, {2 * ks}, {xk}")
        A.e(f"v_lshl_add_u32 {v(KA + ks)}, {t0}, 4, {pi8}")
    A.c("V^T fragment read addresses: row d = 32 dt + r, chunk (2 s + h) ^ ((r >> 1) & 7)")
    yv, r7 = v(X + 7), v(X + 8)
    A.e(f"v_bfe_u32 {t0}, {r}, 1, 3")
    A.e(f"v_xor_b32 {yv}, {h}, {t0}")
    A.e(f"v_lshlrev_b32 {r7}, 7, {r}")
    A.e(f"v_add_u32 {r7}, {V_BASE}, {r7}")
    for si in range(4):
        A.e(f"v_xor_b32 {t0}, {2 * si}, {yv}")
        A.e(f"v_lshl_add_u32 {v(VA + si)}, {t0}, 4, {r7}")
    A.c("K DMA source offsets: piece p of wave w = keys 16 w + 4 p + (lane >> 4), LDS position lane & 15 holds chunk")
    A.c("(lane & 15) ^ (key & 15)")
    l4, l15, key0 = v(X + 7), v(X + 8), v(X + 9)
    A.e(f"v_lshrrev_b32 {l4}, 4, {lane}")
    A.e(f"v_and_b32 {l15}, 15, {lane}")
    A.e(f"v_lshl_add_u32 {key0}, {w}, 4, {l4}")
    for p in range(4):
        A.e(f"v_add_u32 {t0}, {4 * p}, {key0}")
        A.e(f"v_add_u32 {t1}, {4 * p}, {l4}")
        A.e(f"v_xor_b32 {t1}, {l15}, {t1}")
        A.e(f"v_lshlrev_b32 {t1}, 4, {t1}")
        A.e(f"v_lshl_add_u32 {v(KS + p)}, {t0}, 8, {t1}")
    A.c("V^T DMA source offsets: piece p of wave w = rows d = 32 w + 8 p + (lane >> 3), position lane & 7 holds chunk")
    A.c("(lane & 7) ^ ((d >> 1) & 7)")
    l3, l7, d0 = v(X + 7), v(X + 8), v(X + 9)
    A.e(f"v_lshrrev_b32 {l3}, 3, {lane}")
    A.e(f"v_and_b32 {l7}, 7, {lane}")
    A.e(f"v_lshl_add_u32 {d0}, {w}, 5, {l3}")
    for p in range(4):
        A.e(f"v_add_u32 {t0}, {8 * p}, {d0}")
        A.e(f"v_bfe_u32 {t1}, {t0}, 1, 3")
        A.e(f"v_xor_b32 {t1}, {l7}, {t1}")
        A.e(f"v_lshlrev_b32 {t1}, 4, {t1}")
        A.e(f"v_mad_u32_u24 {v(VS + p)}, {t0}, {s(sSP2)}, {t1}")
    A.c("Q / O / lse offsets of this lane's query row w * 64 + r (chain B: + 32 rows)")
    A.e(f"v_lshl_add_u32 {t0}, {w}, 6, {r}")
    A.e(f"v_lshlrev_b32 {t1}, 8, {t0}")
    A.e(f"v_lshl_add_u32 {v(QOFF_A)}, {h}, 4, {t1}")
    A.e(f"v_add_u32 {v(QOFF_B)}, 8192, {v(QOFF_A)}")
    A.e(f"v_mul_lo_u32 {t1}, {t0}, {s(sLDO2)}")
    A.e(f"v_lshl_add_u32 {v(OOFF_A)}, {h}, 4, {t1}")
    A.e(f"s_lshl_b32 {s(sTMP)}, {s(sLDO2)}, 5")
    A.e(f"v_add_u32 {v(OOFF_B)}, {s(sTMP)}, {v(OOFF_A)}")
    A.e(f"v_lshlrev_b32 {v(LOFF)}, 2, {t0}")
    A.c("block loop state; the first block's tiles K(0), V^T(0) -> slot 0, K(1) -> slot 1 and its Q fragments")
    A.e(f"s_mov_b32 {s(sLEFT)}, %[nblk]")
    A.e(f"s_mov_b32 {s(sQT)}, %[qt0]")
    A.e(f"s_mov_b32 {s(sHH)}, %[hh0]")
    A.e(f"s_mov_b32 {s(sBB)}, %[b0]")
    for x in (f"s_mov_b32 {s(sKP)}, {s(sK)}", f"s_mov_b32 {s(sKP + 1)}, {s(sK + 1)}",
              f"s_mov_b32 {s(sVP)}, {s(sV)}", f"s_mov_b32 {s(sVP + 1)}, {s(sV + 1)}",
              f"s_add_u32 {s(sRET)}, {s(sK)}, {SLOT}", f"s_addc_u32 {s(sRET + 1)}, {s(sK + 1)}, 0",
              f"s_mov_b32 {s(sKT)}, {2 * SLOT}", f"s_mov_b32 {s(sVT)}, 128"):
        A.e(x)
    for pc in ([dma_piece("k", p, 0) for p in range(4)] + [dma_piece("v", p, 0) for p in range(4)] +
               [dma_piece("k", p, 1, ptr=sRET) for p in range(4)]):
        A.e(pc[0])
        A.e("s_nop 0")
        A.e(pc[1])
    A.c("Q fragments (B operand of S^T = K Q^T): Q[row][16 ks + 8 h ..]")
    for ch, off in ((CA, QOFF_A), (CB, QOFF_B)):
        for ks in range(8):
            A.e(f"global_load_dwordx4 {ar(ch.Q + 4 * ks, 4)}, {v(off)}, {sr(sQ, 2)} offset:{32 * ks}")
    A.e("s_waitcnt vmcnt(0)")


def block_start(A):
    """Every block: O = 0, the block's first tiles (fetched during the previous block's last iteration, or by the prologue)
    have landed -- the previous block's O / lse stores, the youngest entries of the vector-memory queue, stay in flight --,
    K(0) fragments, S^T of chain A."""
    A.c("================ block start")
    for i in range(128):
        A.e(f"v_accvgpr_write_b32 {a(i)}, 0")
    w16, go = A.new_label("w16"), A.new_label("wgo")
    A.e(f"s_cmp_eq_u64 {sr(sL, 2)}, 0")
    A.e(f"s_cbranch_scc1 {w16}")
    A.e("s_waitcnt vmcnt(18)")                       # 16 O stores + 2 lse stores of the previous block may still be in flight
    A.e(f"s_branch {go}")
    A.label(w16)
    A.e("s_waitcnt vmcnt(16)")
    A.label(go)
    A.e("s_barrier")
    for n in range(16):
        A.e(ds_read_k(n, 0))
    A.e("s_waitcnt lgkmcnt(0)")
    A.e("s_barrier")                                 # every wave holds K(0): slot 0 may be refilled with K(2)
    for n in range(16):
        A.e(mfma_qk(CA, n, first_tile=True))
    A.e("s_nop 7")
    A.e("s_nop 7")


def epilogue(A):
    A.c("================ tail: P V of chain B, last tile")
    for n in range(16):
        A.e(mfma_pv(CB, n))
    A.e("s_nop 7")
    A.e("s_nop 7")
    A.c("================ epilogue: normalise, bf16, 16-byte stores (lane halves exchanged pairwise, T21), LSE")
    for ch, ooff, lrow in ((CA, OOFF_A, 0), (CB, OOFF_B, 32)):
        lt, tmp, inv = v(X), v(X + 1), v(X + 2)
        A.e(f"v_mov_b32 {tmp}, {v(ch.L)}")
        A.e(f"v_mov_b32 {lt}, {v(ch.L)}")
        A.e("s_nop 1")
        A.e(f"v_permlane32_swap_b32 {lt}, {tmp}")
        A.e(f"v_add_f32 {lt}, {lt}, {tmp}")
        A.e(f"v_rcp_f32 {inv}, {lt}")
        A.e(f"v_log_f32 {tmp}, {lt}")
        A.e("s_nop 0")
        A.e(f"v_add_f32 {tmp}, {tmp}, {v(ch.MC)}")
        A.e(f"v_mul_f32 {tmp}, 0x3f317218, {tmp}")              # ln 2: lse = m * scale + ln(l)
        skip = A.new_label("nolse")
        A.e(f"s_cmp_eq_u64 {sr(sL, 2)}, 0")
        A.e(f"s_cbranch_scc1 {skip}")
        A.e("s_mov_b64 exec, 0xffffffff")
        A.e(f"global_store_dword {v(LOFF)}, {tmp}, {sr(sL, 2)} offset:{4 * lrow}")
        A.e("s_mov_b64 exec, -1")
        A.label(skip)
        for dt in range(4):
            for g in (0, 2):
                E = X + 8 + 4 * ((dt * 2 + (g >> 1)) & 3)           # four staging quads in rotation
                rd = [v(X + 24 + j) for j in range(8)]
                for j in range(8):
                    A.e(f"v_accvgpr_read_b32 {rd[j]}, {a(ch.O + 16 * dt + 4 * g + j)}")
                for j in range(8):
                    A.e(f"v_mul_f32 {rd[j]}, {rd[j]}, {inv}")
                for j in range(4):
                    A.e(f"v_cvt_pk_bf16_f32 {v(E + j)}, {rd[2 * j]}, {rd[2 * j + 1]}")
                A.e("s_nop 1")
                A.e(f"v_permlane32_swap_b32 {v(E)}, {v(E + 2)}")
                A.e(f"v_permlane32_swap_b32 {v(E + 1)}, {v(E + 3)}")
                A.e(f"global_store_dwordx4 {v(ooff)}, {vr(E, 4)}, {sr(sO, 2)} offset:{64 * dt + 16 * g}")
    # (no vmcnt(0): the wave ends with its stores in flight, the CU takes the next workgroup meanwhile)


def stamp(A, k):
    """Diagnostic builds only (scratch/fwd64_diag.hip): shader-clock stamp k of this wave into the debug buffer."""
    A.e(f"s_memtime {sr(sRET, 2)}")
    A.e("s_waitcnt lgkmcnt(0)")
    A.e(f"v_mov_b32 {v(DBG_LO)}, {s(sRET)}")
    A.e(f"v_mov_b32 {v(DBG_LO + 1)}, {s(sRET + 1)}")
    A.e("s_mov_b64 exec, 1")
    A.e(f"global_store_dwordx2 {v(DBG_OFF)}, {vr(DBG_LO, 2)}, {sr(sDBG, 2)} offset:{8 * k}")
    A.e("s_mov_b64 exec, -1")


def generate(diag=False, acc=False):
    global ACC
    ACC = acc
    try:
        return _generate(diag)
    finally:
        ACC = False


def _generate(diag):
    A = Asm()
    fixups = []
    if diag:
        A.e(f"s_mov_b32 {s(sDBG)}, %[d_lo]")
        A.e(f"s_mov_b32 {s(sDBG + 1)}, %[d_hi]")
        A.e(f"v_lshrrev_b32 {v(DBG_OFF)}, 6, %[tid]")
        A.e(f"v_lshlrev_b32 {v(DBG_OFF)}, 6, {v(DBG_OFF)}")         # 64 bytes of stamps per wave
        stamp(A, 0)
    prologue(A)
    block = A.new_label("block")
    A.label(block)
    block_start(A)
    if diag:
        stamp(A, 1)
    iteration(A, 0, fixups, first=True)
    if diag:
        stamp(A, 2)
    A.e(f"s_mov_b32 {s(sLOOP)}, {s(sNLOOP)}")
    loop, done = A.new_label("loop"), A.new_label("loopdone")
    A.e(f"s_cmp_eq_u32 {s(sLOOP)}, 0")
    A.e(f"s_cbranch_scc1 {done}")
    A.label(loop)
    iteration(A, 1, fixups)
    iteration(A, 0, fixups)
    A.e(f"s_sub_u32 {s(sLOOP)}, {s(sLOOP)}, 1")
    A.e(f"s_cmp_lg_u32 {s(sLOOP)}, 0")
    A.e(f"s_cbranch_scc1 {loop}")
    A.label(done)
    if diag:
        stamp(A, 3)
    iteration(A, 1, fixups, last=True)
    if diag:
        stamp(A, 4)
    epilogue(A)
    block_advance_stores(A)
    if diag:
        stamp(A, 5)
        A.e(f"s_memrealtime {sr(sRET, 2)}")
        A.e("s_waitcnt lgkmcnt(0)")
        A.e(f"v_mov_b32 {v(DBG_LO)}, {s(sRET)}")
        A.e(f"v_mov_b32 {v(DBG_LO + 1)}, {s(sRET + 1)}")
        A.e("s_mov_b64 exec, 1")
        A.e(f"global_store_dwordx2 {v(DBG_OFF)}, {vr(DBG_LO, 2)}, {sr(sDBG, 2)} offset:48")
        A.e("s_mov_b64 exec, -1")
        A.e("s_waitcnt vmcnt(0)")
    else:
        A.e(f"s_cmp_lg_u32 {s(sLEFT)}, 0")
        A.e(f"s_cbranch_scc1 {block}")
    end = A.new_label("end")
    A.e(f"s_branch {end}")
    A.c("================ out-of-line rescale fix-ups")
    for fix, back, ch in fixups:
        A.label(fix)
        fixup(A, ch, back)
    A.label(end)
    return A.text()


def clobbers(acc=False):
    regs = [f"v{i}" for i in range(4, (255 if acc else V_LAST) + 1)] + [f"a{i}" for i in range(256)] + \
           [f"s{i}" for i in range(S_FIRST, S_LAST + 1)] + ["vcc", "scc", "memory"]
    return ", ".join(f'"{x}"' for x in regs)


HERE = os.path.dirname(os.path.abspath(__file__))
OUT_BODY = os.path.join(HERE, "..", "attn_fwd64_body.inc")
OUT_BODY_Q = os.path.join(HERE, "..", "attn_fwd64q_body.inc")


def render(diag=False, acc=False):
    body = generate(diag, acc)
    name = "ATTN_FWD64Q" if acc else "ATTN_FWD64"
    lines = ["// GENERATED by mixgrpo_amd/csrc/gen/attn_fwd64.py -- do not edit; see that file for the design.",
             f"#define {name}_CLOBBERS " + clobbers(acc),
             f"#define {name}_BODY \\"]
    for ln in body.rstrip("\n").split("\n"):
        lines.append('  "' + ln.replace("\\", "\\\\").replace('"', '\\"') + '\\n" \\')
    lines.append('  ""')
    return "\n".join(lines) + "\n"


def write():
    for path, acc in ((OUT_BODY, False), (OUT_BODY_Q, True)):
        txt = render(acc=acc)
        old = open(path).read() if os.path.exists(path) else None
        if old != txt:
            with open(path, "w") as f:
                f.write(txt)
    return OUT_BODY, OUT_BODY_Q


if __name__ == "__main__":
    if "--print" in sys.argv:
        sys.stdout.write(generate("--diag" in sys.argv, "--acc" in sys.argv))
    elif "--diag" in sys.argv:                      # scratch/fwd64_diag.hip includes this one
        for a_ in sys.argv:
            if a_.startswith("--timing-only="):
                TIMING_ONLY.update(a_.split("=")[1].split(","))
        out = os.path.join(HERE, "..", "..", "..", "scratch", "attn_fwd64_diag_body.inc")
        with open(out, "w") as f:
            f.write(render(diag=True, acc="--acc" in sys.argv))
        print(out)
    else:
        print(write())
