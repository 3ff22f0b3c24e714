#!/usr/bin/env python3
"""Generator of the hand-placed gfx950 instruction stream of `attn_bwd_dkv64_kernel` (mixgrpo_amd/csrc/attention_bwd.hip).

dK and dV of the joint attention backward (autograd of F.scaled_dot_product_attention at the reference's call site
fastvideo/train_grpo_flux.py:134-144), for S % 256 == 0; the 8-wave kernel stays for other shapes.

One wave per SIMD; a wave owns 64 KEYS = chains a and b of 32 and keeps dV^T and dK^T of both in all 256 accumulator
registers; the workgroup (4 waves = 256 keys) sweeps 32-query blocks.  Every Q / dO / dO^T / Q^T fragment read from LDS feeds
both chains (the 8-wave kernel: one).  With the accumulators taking half the register file the chains cannot be staggered
like the forward's (a block's fragments would have to be held across segments), so they run in lockstep and the vector work
of block i runs under the ACCUMULATE products of block i-1.  Iteration i = 64 MFMA gaps:

    gaps  0..15  A(i):   S'_c  = Q(i) K_c^T - lse / scale   (row constants as the chains' initial accumulators)
    gaps 16..31          dP'_c = dO(i) V_c^T - delta         (V fragments from a wave-private LDS area)
    gaps 32..47  B(i-1): dV_c^T += dO^T(i-1) P_c(i-1)        | VALU(i): p = exp2(c S') in place
    gaps 48..63          dK_c^T += Q^T(i-1) dS_c(i-1)        |          P(i) -> bf16 pairs, ds = p dP' in place
    gaps  2.. 9 of the NEXT iteration: dS(i) -> bf16 pairs (dS(i-1) is dead, dP'(i) not yet overwritten)
  so P and dS need ONE buffer each.  dK is scaled by `scale` in the epilogue (dS carries no scale).

Registers: dV_a a[0:63], dV_b a[64:127], dK_a a[128:191], dK_b a[192:255]; K fragments v[4:67]; S' v[68:99], dP' v[100:131],
P v[132:147], dS v[148:163], row constants v[164:179], a ring of 12 fragment registers v[180:227].
LDS (146 KiB): Q | dO row-major tiles, 3 slots (read addresses rotate in a register: no unrolling by 3); Q^T | dO^T tiles,
2 slots (the loop is unrolled by 2); lse | delta rows, 4 slots; the waves' V fragments (4 x 16 KiB).  Tiles are fetched two
iterations ahead (Q | dO, lse | delta) so that the first fragments of an iteration are read before the barrier in front of it.
The `s_waitcnt lgkmcnt` in front of each MFMA is derived from the in-order LDS queue by the generator.
Checked on the CPU by tests/test_attn_bwd64_emulated.py (interpreter + hazard pass) before it runs on a GPU.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from attn_fwd64 import Asm, a, ar, s, sr, v, vr  # noqa: E402

MFMA = "v_mfma_f32_32x32x16_bf16"

# ------------------------------------------------------------------------------------------------ register map
DV_A, DV_B, DK_A, DK_B = 0, 64, 128, 192
KF_A, KF_B = 4, 36
S_A, S_B, DP_A, DP_B = 68, 84, 100, 116
P_A, P_B, DS_A, DS_B = 132, 140, 148, 156
C = 164
RING = 180                       # 12 fragment slots
QB, QB0, VPRIV, TA, CB, CB0 = 228, 229, 230, 231, 233, 234      # TA: two registers (s2 = 0, 1)
QSRC, DOSRC, TSRC, CSRC = 235, 237, 239, 241                     # 2, 2, 2 registers and a 64-bit pair
T = 243                          # 8 Q / dO fragment addresses of the coming A phase
X = 252                          # v[252:255]: short-lived scratch (even: 64-bit address pairs)
V_LAST = 255

sW, sWOFF2 = 60, 61              # wave id, w * 2048
sQP, sDOP, sQTP, sDOTP = 62, 64, 66, 68       # pairs: DMA base pointers
sLOOP, sQROW, sTT, sTMP, sQRD, sQDST, sCRD, sCDST, sCOFF = 70, 71, 72, 73, 74, 75, 76, 77, 78
sQ, sDO, sQT, sDOT = 80, 82, 84, 86           # tensor bases (pairs)
sSP2, sLDO2, sCS, sSCALE, sNIS, sNLOOP, sQMAX, sLDO32, sCMAX = 88, 89, 90, 91, 92, 93, 94, 95, 96
sDK, sDV = 62, 64                # epilogue: the DMA pointers' registers
S_FIRST, S_LAST = 60, 97

QDO_SLOT = 16384
T_BASE, C_BASE, VP_BASE = 49152, 81920, 83968
LDS_BYTES = VP_BASE + 4 * 16384
LEAD = 5                         # gaps between a fragment's ds_read and its first MFMA


class Chain:
    def __init__(self, name, DV, DK, KF, S, DP, P, DS):
        self.name, self.DV, self.DK, self.KF, self.S, self.DP, self.P, self.DS = name, DV, DK, KF, S, DP, P, DS


CHAINS = (Chain("a", DV_A, DK_A, KF_A, S_A, DP_A, P_A, DS_A), Chain("b", DV_B, DK_B, KF_B, S_B, DP_B, P_B, DS_B))


def ring(i):
    return vr(RING + 4 * i, 4)


# fragment id -> (ring slot, first-use gap)
PLAN = {}
for _ks in range(8):
    PLAN[("q", _ks)] = (_ks % 4, 2 * _ks)
    PLAN[("do", _ks)] = (_ks % 4, 16 + 2 * _ks)
    PLAN[("va", _ks)] = (4 + _ks % 4, 16 + 2 * _ks)
    PLAN[("vb", _ks)] = (8 + _ks % 4, 17 + 2 * _ks)
for _n in range(8):
    PLAN[("dot", _n)] = (4 + _n % 4, 32 + 2 * _n)
    PLAN[("qt", _n)] = (8 + _n % 4, 48 + 2 * _n)


def read_instr(fid, tslot):
    """ds_read of a fragment; tslot = slot of the Q^T | dO^T ring the B phase reads."""
    kind, i = fid
    dst = ring(PLAN[fid][0])
    if kind == "q":
        return f"ds_read_b128 {dst}, {v(T + i)}"
    if kind == "do":
        return f"ds_read_b128 {dst}, {v(T + i)} offset:8192"
    if kind == "va":
        return f"ds_read_b128 {dst}, {v(VPRIV)} offset:{i * 1024}"
    if kind == "vb":
        return f"ds_read_b128 {dst}, {v(VPRIV)} offset:{8192 + i * 1024}"
    s2, dt = i >> 2, i & 3
    off = tslot * 16384 + dt * 2048 + (8192 if kind == "dot" else 0)
    return f"ds_read_b128 {dst}, {v(TA + s2)} offset:{off}"


def c_reads(which):
    """Four ds_read_b128 of this lane half's 16 row constants: lse (0) or delta (1) of rows 8h..8h+7 and 16+8h..16+8h+7."""
    return [f"ds_read_b128 {vr(C + 4 * i, 4)}, {v(CB)} offset:{128 * which + off}" for i, off in enumerate((0, 16, 64, 80))]


def mfma_text(g):
    """(instruction, fragment ids it needs) of gap g."""
    ch = CHAINS[g & 1]
    if g < 16:
        ks = g >> 1
        srcc = vr(C, 16) if ks == 0 else vr(ch.S, 16)
        return (f"{MFMA} {vr(ch.S, 16)}, {ring(PLAN[('q', ks)][0])}, {vr(ch.KF + 4 * ks, 4)}, {srcc}",
                [("q", ks)] + ([("c0",)] if ks == 0 else []))
    if g < 32:
        ks = (g - 16) >> 1
        vf = ("v" + ch.name, ks)
        srcc = vr(C, 16) if ks == 0 else vr(ch.DP, 16)
        return (f"{MFMA} {vr(ch.DP, 16)}, {ring(PLAN[('do', ks)][0])}, {ring(PLAN[vf][0])}, {srcc}",
                [("do", ks), vf] + ([("c1",)] if ks == 0 else []))
    if g < 48:
        n = (g - 32) >> 1
        o = ar(ch.DV + 16 * (n & 3), 16)
        return f"{MFMA} {o}, {ring(PLAN[('dot', n)][0])}, {vr(ch.P + 4 * (n >> 2), 4)}, {o}", [("dot", n)]
    n = (g - 48) >> 1
    o = ar(ch.DK + 16 * (n & 3), 16)
    return f"{MFMA} {o}, {ring(PLAN[('qt', n)][0])}, {vr(ch.DS + 4 * (n >> 2), 4)}, {o}", [("qt", n)]


class LdsQueue:
    """The wave's in-order LDS queue, to derive `s_waitcnt lgkmcnt(N)`."""

    def __init__(self, A):
        self.A, self.issued, self.done = A, [], 0

    def issue(self, fid, instr):
        self.A.e(instr)
        self.issued.append(fid)

    def preload(self, fids):
        self.issued.extend(fids)

    def need(self, fids):
        idx = max(len(self.issued) - self.issued[::-1].index(f) for f in fids)      # position after the LAST issue of f
        if idx > self.done:
            out = len(self.issued) - idx
            assert out <= 15, f"lgkmcnt({out}) is not encodable"
            self.A.e(f"s_waitcnt lgkmcnt({out})")
            self.done = idx


def valu_of_gap(g, first, last):
    """VALU instructions of gap g of the steady-state iteration (see the module docstring)."""
    out = []

    def sreg(e):
        return v(CHAINS[e >> 4].S + (e & 15))

    if 2 <= g <= 9 and not first:                      # dS(i-1) pairs out of the dP' registers
        for k in (2 * (g - 2), 2 * (g - 2) + 1):
            ch, kk = CHAINS[k >> 3], k & 7
            out.append(f"v_cvt_pk_bf16_f32 {v(ch.DS + kk)}, {v(ch.DP + 2 * kk)}, {v(ch.DP + 2 * kk + 1)}")
    if 10 <= g <= 13:                                  # C1 = -delta (read at gaps 4, 5)
        for i in range(4 * (g - 10), 4 * (g - 10) + 4):
            out.append(f"v_xor_b32 {v(C + i)}, 0x80000000, {v(C + i)}")
    if 32 <= g <= 48:                                  # p = exp2(c S') in place: two elements per gap, exp one gap behind
        j = g - 32
        if j < 16:
            out.append(f"v_mul_f32 {sreg(2 * j)}, {sreg(2 * j)}, {s(sCS)}")
            out.append(f"v_mul_f32 {sreg(2 * j + 1)}, {sreg(2 * j + 1)}, {s(sCS)}")
        if j >= 1:
            out.append(f"v_exp_f32 {sreg(2 * j - 2)}, {sreg(2 * j - 2)}")
            out.append(f"v_exp_f32 {sreg(2 * j - 1)}, {sreg(2 * j - 1)}")
    if 49 <= g <= 63:                                  # P(i) pairs and ds = p dP' in place (16 pairs over 15 gaps)
        ks_ = [g - 49] + ([15] if g == 63 else [])
        for k in ks_:
            ch, kk = CHAINS[k >> 3], k & 7
            out.append(f"v_cvt_pk_bf16_f32 {v(ch.P + kk)}, {v(ch.S + 2 * kk)}, {v(ch.S + 2 * kk + 1)}")
            out.append(f"v_mul_f32 {v(ch.DP + 2 * kk)}, {v(ch.S + 2 * kk)}, {v(ch.DP + 2 * kk)}")
            out.append(f"v_mul_f32 {v(ch.DP + 2 * kk + 1)}, {v(ch.S + 2 * kk + 1)}, {v(ch.DP + 2 * kk + 1)}")
    if not last:
        if 44 <= g <= 47:                              # C0 of block i+1 (read at gaps 36..39) = -lse / scale
            for i in range(4 * (g - 44), 4 * (g - 44) + 4):
                out.append(f"v_mul_f32 {v(C + i)}, {v(C + i)}, {s(sNIS)}")
        if g == 48:                                    # Q | dO read addresses of block i+1 (slot rotation)
            out.append(f"v_add_u32 {v(QB)}, {s(sQRD)}, {v(QB0)}")
        if 49 <= g <= 56:
            out.append(f"v_xor_b32 {v(T + g - 49)}, {32 * (g - 49)}, {v(QB)}")
    return out


def rotate(reg, step, limit, base=None):
    """reg = base + ((reg - base + step) mod limit) in scalar code."""
    out = []
    if base is not None:
        out.append(f"s_sub_u32 {s(reg)}, {s(reg)}, {base}")
    out += [f"s_add_u32 {s(reg)}, {s(reg)}, {step}", f"s_cmp_ge_u32 {s(reg)}, {limit}",
            f"s_cselect_b32 {s(sTMP)}, {limit}, 0", f"s_sub_u32 {s(reg)}, {s(reg)}, {s(sTMP)}"]
    if base is not None:
        out.append(f"s_add_u32 {s(reg)}, {s(reg)}, {base}")
    return out


def iteration_setup(last):
    """Scalar set-up at the top of iteration i: DMA pointers of Q | dO (i+2) [clamped], Q^T | dO^T (i); slot rotation."""
    out = []
    if not last:
        out += [f"s_min_u32 {s(sTMP)}, {s(sQROW)}, {s(sQMAX)}",                # Q byte offset of block i+2 (8192 per block)
                f"s_add_u32 {s(sQP)}, {s(sQ)}, {s(sTMP)}", f"s_addc_u32 {s(sQP + 1)}, {s(sQ + 1)}, 0",
                f"s_lshr_b32 {s(sTMP)}, {s(sTMP)}, 13", f"s_mul_i32 {s(sTMP)}, {s(sTMP)}, {s(sLDO32)}",
                f"s_add_u32 {s(sDOP)}, {s(sDO)}, {s(sTMP)}", f"s_addc_u32 {s(sDOP + 1)}, {s(sDO + 1)}, 0",
                f"s_add_u32 {s(sQROW)}, {s(sQROW)}, 8192"]
    out += [f"s_add_u32 {s(sQTP)}, {s(sQT)}, {s(sTT)}", f"s_addc_u32 {s(sQTP + 1)}, {s(sQT + 1)}, 0",
            f"s_add_u32 {s(sDOTP)}, {s(sDOT)}, {s(sTT)}", f"s_addc_u32 {s(sDOTP + 1)}, {s(sDOT + 1)}, 0",
            f"s_add_u32 {s(sTT)}, {s(sTT)}, 64"]
    return out


def dma_list(par, last):
    """(gap, m0 write, load) of the iteration's LDS-DMA pieces: Q, dO of block i+2; Q^T, dO^T of block i (slot par)."""
    out = []
    g = 12
    if not last:
        for kind, off, ptr, src in (("q", 0, sQP, QSRC), ("do", 8192, sDOP, DOSRC)):
            for p in range(2):
                out.append((g, f"s_add_i32 m0, {s(sQDST)}, {off + p * 1024}",
                            f"global_load_lds_dwordx4 {v(src + p)}, {sr(ptr, 2)}"))
                g += 2
    for kind, off, ptr in (("qt", 0, sQTP), ("dot", 8192, sDOTP)):
        for p in range(2):
            out.append((g, f"s_add_i32 m0, {s(sWOFF2)}, {T_BASE + par * 16384 + off + p * 1024}",
                        f"global_load_lds_dwordx4 {v(TSRC + p)}, {sr(ptr, 2)}"))
            g += 2
    return out


def c_dma(A):
    """lse | delta of block i+2 (32 + 32 floats) into the rotating 256-byte slot: one dword DMA, wave 0 only."""
    skip = A.new_label("noc")
    A.e(f"s_cmp_lg_u32 {s(sW)}, 0")
    A.e(f"s_cbranch_scc1 {skip}")
    A.e(f"s_min_u32 {s(sTMP)}, {s(sCOFF)}, {s(sCMAX)}")
    A.e(f"v_add_co_u32 {v(X)}, vcc, {s(sTMP)}, {v(CSRC)}")
    A.e(f"v_addc_co_u32 {v(X + 1)}, vcc, 0, {v(CSRC + 1)}, vcc")
    A.e(f"s_mov_b32 m0, {s(sCDST)}")
    A.e("s_nop 0")
    A.e(f"global_load_lds_dword {vr(X, 2)}, off")
    A.label(skip)
    A.e(f"s_add_u32 {s(sCOFF)}, {s(sCOFF)}, 128")


def emit_iteration(A, par, first=False, last=False):
    """Iteration i (i & 1 == par): A(i), then B(i-1) under VALU(i) [first: no B(i-1)]."""
    A.c(f"================ iteration parity {par}{' FIRST' if first else ''}{' LAST' if last else ''}")
    Q = LdsQueue(A)
    # issued at the end of the previous iteration (or of the prologue), in this order: C0, then the fragments used before LEAD
    pre = sorted((f for f in PLAN if PLAN[f][1] - LEAD < 0), key=lambda f: PLAN[f][1])
    Q.preload([("c0",)] * 4 + pre)
    for x in iteration_setup(last):
        A.e(x)
    for x in rotate(sQDST, QDO_SLOT, 3 * QDO_SLOT):    # Q | dO destination slot (i+2) % 3; the wave offset is added per piece
        A.e(x)
    dmas = {g: (m0w, ld) for g, m0w, ld in dma_list(par, last)}
    reads_at = [[] for _ in range(64)]
    for fid, (slot, use) in sorted(PLAN.items(), key=lambda kv: kv[1][1]):
        if use - LEAD >= 0 and not (first and fid[0] in ("dot", "qt")):
            reads_at[use - LEAD].append((fid, read_instr(fid, 1 - par)))
    for i, ins in enumerate(c_reads(1)):
        reads_at[4 + (i >> 1)].append((("c1",), ins))
    if not last:
        for i, ins in enumerate(c_reads(0)):
            reads_at[36 + i].append((("c0n",), ins))
    for g in range(64):
        text, needs = mfma_text(g)
        is_b = g >= 32
        if not (first and is_b):
            Q.need(needs)
            A.e(text)
        if g == 10:
            Q.need([("c1",)])
        if g == 44 and not last:
            Q.need([("c0n",)])
        fill = valu_of_gap(g, first, last)
        if g in dmas:
            m0w, ld = dmas[g]
            if m0w.startswith(f"s_add_i32 m0, {s(sQDST)}"):    # Q | dO pieces: m0 = rotating slot + w * 2048 + piece
                A.e(f"s_add_u32 {s(sTMP)}, {s(sQDST)}, {s(sWOFF2)}")
                m0w = m0w.replace(s(sQDST), s(sTMP))
            A.e(m0w)
            A.e(fill.pop(0) if fill else "s_nop 0")
            A.e(ld)
        for x in fill:
            A.e(x)
        if g == 30 and not last:
            for x in rotate(sCDST, 256, 1024, base=C_BASE):
                A.e(x)
            c_dma(A)
        if g == 34 and not last:
            for x in rotate(sCRD, 256, 1024) + rotate(sQRD, QDO_SLOT, 3 * QDO_SLOT):
                A.e(x)
            A.e(f"v_add_u32 {v(CB)}, {s(sCRD)}, {v(CB0)}")
        for fid, ins in reads_at[g]:
            Q.issue(fid, ins)
    if not last:
        for fid in pre:                                # block i+1's first fragments (its tiles landed an iteration ago)
            A.e(read_instr(fid, 0))
    A.e("s_waitcnt vmcnt(0)")
    A.e("s_barrier")


# ------------------------------------------------------------------------------------------------ prologue / epilogue
def prologue(A):
    for nm, reg in (("q", sQ), ("do", sDO), ("qt", sQT), ("dot", sDOT)):
        A.e(f"s_mov_b32 {s(reg)}, %[{nm}_lo]")
        A.e(f"s_mov_b32 {s(reg + 1)}, %[{nm}_hi]")
    for dst, nm in ((sSP2, "sp2"), (sLDO2, "ldo2"), (sCS, "cs"), (sSCALE, "scale"), (sNIS, "nis"), (sNLOOP, "nloop"),
                    (sQMAX, "qmax"), (sLDO32, "ldo32"), (sCMAX, "cmax")):
        A.e(f"s_mov_b32 {s(dst)}, %[{nm}]")
    lane, w, r, h = v(RING), v(RING + 1), v(RING + 2), v(RING + 3)       # the ring is free in the prologue
    t0, t1, t2, t3 = v(RING + 4), v(RING + 5), v(RING + 6), v(RING + 7)
    A.e(f"v_and_b32 {lane}, 63, %[tid]")
    A.e(f"v_lshrrev_b32 {w}, 6, %[tid]")
    A.e(f"v_and_b32 {r}, 31, {lane}")
    A.e(f"v_lshrrev_b32 {h}, 5, {lane}")
    A.e(f"v_readfirstlane_b32 {s(sW)}, {w}")
    A.e(f"s_lshl_b32 {s(sWOFF2)}, {s(sW)}, 11")
    A.c("Q | dO fragment reads: MFMA row r reads tile row pi(r) (bits 2, 3 exchanged), chunk (2 ks + h) ^ (row & 15)")
    pi = v(RING + 8)
    A.e(f"v_and_b32 {t0}, 0x13, {r}")
    A.e(f"v_and_b32 {t1}, 4, {r}")
    A.e(f"v_lshlrev_b32 {t1}, 1, {t1}")
    A.e(f"v_and_b32 {t2}, 8, {r}")
    A.e(f"v_lshrrev_b32 {t2}, 1, {t2}")
    A.e(f"v_or3_b32 {pi}, {t0}, {t1}, {t2}")
    A.e(f"v_and_b32 {t0}, 15, {pi}")
    A.e(f"v_xor_b32 {t0}, {h}, {t0}")
    A.e(f"v_lshlrev_b32 {t1}, 8, {pi}")
    A.e(f"v_lshl_add_u32 {v(QB0)}, {t0}, 4, {t1}")
    A.c("V fragments, wave-private: [chain][ks][lane] 16-byte pieces")
    A.e(f"v_lshlrev_b32 {t0}, 4, {lane}")
    A.e(f"v_lshl_add_u32 {v(VPRIV)}, {w}, 14, {t0}")
    A.e(f"v_add_u32 {v(VPRIV)}, {VP_BASE}, {v(VPRIV)}")
    A.c("Q^T | dO^T fragment reads: row d = 32 dt + r of the [128][32] image (64-byte rows), chunk (2 s2 + h) ^ ((r >> 2) & 3)")
    A.e(f"v_bfe_u32 {t0}, {r}, 2, 2")
    A.e(f"v_xor_b32 {t0}, {h}, {t0}")                                      # h ^ f(r): bit 0 of the chunk index
    A.e(f"v_lshlrev_b32 {t1}, 6, {r}")
    A.e(f"v_add_u32 {t1}, {T_BASE}, {t1}")
    for s2 in range(2):
        A.e(f"v_xor_b32 {t2}, {2 * s2}, {t0}")
        A.e(f"v_lshl_add_u32 {v(TA + s2)}, {t2}, 4, {t1}")
    A.c("row constants: this lane half reads rows 8 h .. and 16 + 8 h .. of the 32-float lse | delta rows")
    A.e(f"v_lshlrev_b32 {v(CB0)}, 5, {h}")
    A.e(f"v_add_u32 {v(CB0)}, {C_BASE}, {v(CB0)}")
    A.c("DMA sources.  Q / dO: piece p of wave w = tile rows 8 w + 4 p + (lane >> 4), position lane & 15 holds chunk ^ (row & 15)")
    l4, l15 = v(RING + 8), v(RING + 9)
    A.e(f"v_lshrrev_b32 {l4}, 4, {lane}")
    A.e(f"v_and_b32 {l15}, 15, {lane}")
    for p in range(2):
        A.e(f"v_lshl_add_u32 {t0}, {w}, 3, {l4}")
        A.e(f"v_add_u32 {t0}, {4 * p}, {t0}")                              # row
        A.e(f"v_and_b32 {t1}, 15, {t0}")
        A.e(f"v_xor_b32 {t1}, {l15}, {t1}")
        A.e(f"v_lshlrev_b32 {t1}, 4, {t1}")
        A.e(f"v_lshl_add_u32 {v(QSRC + p)}, {t0}, 8, {t1}")
        A.e(f"v_mad_u32_u24 {v(DOSRC + p)}, {t0}, {s(sLDO2)}, {t1}")
    A.c("Q^T / dO^T: piece p of wave w = rows d = 32 w + 16 p + (lane >> 2), position lane & 3 holds chunk ^ ((d >> 2) & 3)")
    l2, l3 = v(RING + 8), v(RING + 9)
    A.e(f"v_lshrrev_b32 {l2}, 2, {lane}")
    A.e(f"v_and_b32 {l3}, 3, {lane}")
    for p in range(2):
        A.e(f"v_lshl_add_u32 {t0}, {w}, 5, {l2}")
        A.e(f"v_add_u32 {t0}, {16 * p}, {t0}")                             # d
        A.e(f"v_bfe_u32 {t1}, {t0}, 2, 2")
        A.e(f"v_xor_b32 {t1}, {l3}, {t1}")
        A.e(f"v_lshlrev_b32 {t1}, 4, {t1}")
        A.e(f"v_mad_u32_u24 {v(TSRC + p)}, {t0}, {s(sSP2)}, {t1}")
    A.c("lse | delta: lanes 0..31 the block's lse, lanes 32..63 its delta (64-bit per-lane addresses)")
    A.e(f"v_lshlrev_b32 {t0}, 2, {r}")
    A.e(f"v_mov_b32 {t1}, %[lse_lo]")
    A.e(f"v_mov_b32 {t2}, %[lse_hi]")
    A.e(f"v_mov_b32 {t3}, %[dl_lo]")
    A.e(f"v_cmp_lt_u32 vcc, 31, {lane}")
    A.e(f"v_cndmask_b32 {t1}, {t1}, {t3}, vcc")
    A.e(f"v_mov_b32 {t3}, %[dl_hi]")
    A.e(f"v_cndmask_b32 {t2}, {t2}, {t3}, vcc")
    A.e(f"v_add_co_u32 {v(CSRC)}, vcc, {t0}, {t1}")
    A.e(f"v_addc_co_u32 {v(CSRC + 1)}, vcc, 0, {t2}, vcc")
    A.c("K fragments (B operands) and the V fragments' DMA: row 32 c + r of this wave's 64 keys, chunk 2 ks + h")
    koff = v(RING + 10)
    A.e(f"v_lshl_add_u32 {t0}, {w}, 6, {r}")
    A.e(f"v_lshlrev_b32 {t0}, 8, {t0}")
    A.e(f"v_lshl_add_u32 {koff}, {h}, 4, {t0}")
    A.e(f"s_mov_b32 {s(sQP)}, %[k_lo]")
    A.e(f"s_mov_b32 {s(sQP + 1)}, %[k_hi]")
    A.e(f"s_mov_b32 {s(sDOP)}, %[v_lo]")
    A.e(f"s_mov_b32 {s(sDOP + 1)}, %[v_hi]")
    A.e(f"v_add_u32 {v(RING + 11)}, 8192, {koff}")
    for ci, ch in enumerate(CHAINS):
        for ks in range(8):
            A.e(f"global_load_dwordx4 {vr(ch.KF + 4 * ks, 4)}, {v(RING + 10 + ci)}, {sr(sQP, 2)} offset:{32 * ks}")
    A.e(f"s_lshl_b32 {s(sTMP)}, {s(sW)}, 14")
    A.e(f"s_add_u32 {s(sTMP)}, {s(sTMP)}, {VP_BASE}")
    for ci in range(2):
        for ks in range(8):
            # (an LDS-DMA's instruction offset is added to the LDS address too: m0 takes it back out)
            A.e(f"s_add_i32 m0, {s(sTMP)}, {ci * 8192 + ks * 1024 - 32 * ks}")
            A.e("s_nop 0")
            A.e(f"global_load_lds_dwordx4 {v(RING + 10 + ci)}, {sr(sDOP, 2)} offset:{32 * ks}")
    A.c("first tiles: Q | dO of blocks 0, 1 -> slots 0, 1; lse | delta of blocks 0, 1 -> slots 0, 1")
    A.e(f"s_mov_b32 {s(sQROW)}, 0")
    A.e(f"s_mov_b32 {s(sTT)}, 0")
    A.e(f"s_mov_b32 {s(sCOFF)}, 0")
    A.e(f"s_mov_b32 {s(sCDST)}, {C_BASE - 256}")
    A.e(f"s_sub_u32 {s(sQDST)}, 0, {QDO_SLOT}")
    for blk in range(2):
        for x in iteration_setup(False)[:8]:
            A.e(x)
        A.e(f"s_add_u32 {s(sQDST)}, {s(sQDST)}, {QDO_SLOT}")
        A.e(f"s_add_u32 {s(sTMP)}, {s(sQDST)}, {s(sWOFF2)}")
        for off, ptr, src in ((0, sQP, QSRC), (8192, sDOP, DOSRC)):
            for p in range(2):
                A.e(f"s_add_i32 m0, {s(sTMP)}, {off + p * 1024}")
                A.e("s_nop 0")
                A.e(f"global_load_lds_dwordx4 {v(src + p)}, {sr(ptr, 2)}")
        A.e(f"s_add_u32 {s(sCDST)}, {s(sCDST)}, 256")
        c_dma(A)
    A.e(f"s_mov_b32 {s(sQRD)}, 0")
    A.e(f"s_mov_b32 {s(sCRD)}, 0")
    A.c("accumulators = 0")
    for i in range(256):
        A.e(f"v_accvgpr_write_b32 {a(i)}, 0")
    A.e("s_waitcnt vmcnt(0)")
    A.e("s_barrier")
    A.c("what an iteration leaves for the next one: read addresses, C0, the first fragments -- here for block 0")
    A.e(f"v_mov_b32 {v(CB)}, {v(CB0)}")
    A.e(f"v_mov_b32 {v(QB)}, {v(QB0)}")
    for ks in range(8):
        A.e(f"v_xor_b32 {v(T + ks)}, {32 * ks}, {v(QB)}")
    for ins in c_reads(0):
        A.e(ins)
    A.e("s_waitcnt lgkmcnt(0)")
    for i in range(16):
        A.e(f"v_mul_f32 {v(C + i)}, {v(C + i)}, {s(sNIS)}")


def epilogue(A):
    A.c("================ tail: dS of the last block, then B(last) alone")
    for k in range(16):
        ch, kk = CHAINS[k >> 3], k & 7
        A.e(f"v_cvt_pk_bf16_f32 {v(ch.DS + kk)}, {v(ch.DP + 2 * kk)}, {v(ch.DP + 2 * kk + 1)}")
    for n in range(8):
        A.e(read_instr(("dot", n), 1))
        A.e(read_instr(("qt", n), 1))
        if n % 4 == 3:
            A.e("s_waitcnt lgkmcnt(0)")
            for m in range(n - 3, n + 1):
                for g in (32 + 2 * m, 33 + 2 * m, 48 + 2 * m, 49 + 2 * m):
                    A.e(mfma_text(g)[0])
    A.e("s_nop 7")
    A.e("s_nop 7")
    A.c("================ epilogue: dV, dK * scale -> bf16, 16-byte stores (lane halves exchanged pairwise)")
    lane, w, r, h, t0 = v(RING), v(RING + 1), v(RING + 2), v(RING + 3), v(RING + 4)
    off_a, off_b = v(RING + 5), v(RING + 6)
    A.e(f"v_and_b32 {lane}, 63, %[tid]")
    A.e(f"v_lshrrev_b32 {w}, 6, %[tid]")
    A.e(f"v_and_b32 {r}, 31, {lane}")
    A.e(f"v_lshrrev_b32 {h}, 5, {lane}")
    A.e(f"v_lshl_add_u32 {t0}, {w}, 6, {r}")
    A.e(f"v_lshlrev_b32 {t0}, 8, {t0}")
    A.e(f"v_lshl_add_u32 {off_a}, {h}, 4, {t0}")
    A.e(f"v_add_u32 {off_b}, 8192, {off_a}")
    A.e(f"s_mov_b32 {s(sDK)}, %[dk_lo]")
    A.e(f"s_mov_b32 {s(sDK + 1)}, %[dk_hi]")
    A.e(f"s_mov_b32 {s(sDV)}, %[dv_lo]")
    A.e(f"s_mov_b32 {s(sDV + 1)}, %[dv_hi]")
    E0 = RING + 8                                        # 4 staging quads RING+8 .. RING+23, 8 read registers RING+24 .. +31
    cnt = 0
    for acc, ptr, scaled in ((DV_A, sDV, False), (DV_B, sDV, False), (DK_A, sDK, True), (DK_B, sDK, True)):
        off = off_a if acc in (DV_A, DK_A) else off_b
        for dt in range(4):
            for g in (0, 2):
                E = E0 + 4 * (cnt & 3)
                cnt += 1
                rd = [v(RING + 24 + j) for j in range(8)]
                for j in range(8):
                    A.e(f"v_accvgpr_read_b32 {rd[j]}, {a(acc + 16 * dt + 4 * g + j)}")
                if scaled:
                    for j in range(8):
                        A.e(f"v_mul_f32 {rd[j]}, {rd[j]}, {s(sSCALE)}")
                for j in range(4):
                    A.e(f"v_cvt_pk_bf16_f32 {v(E + j)}, {rd[2 * j]}, {rd[2 * j + 1]}")
                A.e("s_nop 1")
                A.e(f"v_permlane32_swap_b32 {v(E)}, {v(E + 2)}")
                A.e(f"v_permlane32_swap_b32 {v(E + 1)}, {v(E + 3)}")
                A.e(f"global_store_dwordx4 {off}, {vr(E, 4)}, {sr(ptr, 2)} offset:{64 * dt + 16 * g}")


def generate():
    A = Asm()
    prologue(A)
    for fid in sorted((f for f in PLAN if PLAN[f][1] - LEAD < 0), key=lambda f: PLAN[f][1]):
        A.e(read_instr(fid, 0))
    emit_iteration(A, 0, first=True)
    A.e(f"s_mov_b32 {s(sLOOP)}, {s(sNLOOP)}")
    loop, done = A.new_label("loop"), A.new_label("loopdone")
    A.e(f"s_cmp_eq_u32 {s(sLOOP)}, 0")
    A.e(f"s_cbranch_scc1 {done}")
    A.label(loop)
    emit_iteration(A, 1)
    emit_iteration(A, 0)
    A.e(f"s_sub_u32 {s(sLOOP)}, {s(sLOOP)}, 1")
    A.e(f"s_cmp_lg_u32 {s(sLOOP)}, 0")
    A.e(f"s_cbranch_scc1 {loop}")
    A.label(done)
    emit_iteration(A, 1, last=True)
    epilogue(A)
    return A.text()


def clobbers():
    regs = [f"v{i}" for i in range(4, V_LAST + 1)] + [f"a{i}" for i in range(256)] + \
           [f"s{i}" for i in range(S_FIRST, S_LAST + 1)] + ["vcc", "scc", "memory"]
    return ", ".join(f'"{x}"' for x in regs)


HERE = os.path.dirname(os.path.abspath(__file__))
OUT_BODY = os.path.join(HERE, "..", "attn_bwd_dkv64_body.inc")


def render():
    body = generate()
    lines = ["// GENERATED by mixgrpo_amd/csrc/gen/attn_bwd_dkv64.py -- do not edit; see that file for the design.",
             "#define ATTN_BWD_DKV64_CLOBBERS " + clobbers(),
             "#define ATTN_BWD_DKV64_BODY \\"]
    for ln in body.rstrip("\n").split("\n"):
        lines.append('  "' + ln.replace("\\", "\\\\").replace('"', '\\"') + '\\n" \\')
    lines.append('  ""')
    return "\n".join(lines) + "\n"


def write(path=OUT_BODY):
    txt = render()
    old = open(path).read() if os.path.exists(path) else None
    if old != txt:
        with open(path, "w") as f:
            f.write(txt)
    return path


if __name__ == "__main__":
    if "--print" in sys.argv:
        sys.stdout.write(generate())
    else:
        print(write())
