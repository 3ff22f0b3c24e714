#!/usr/bin/env python3
"""Generator of the hand-placed gfx950 instruction stream of `attn_bwd_dq64_kernel` (mixgrpo_amd/csrc/attention_bwd.hip).

dQ of the joint attention backward (autograd of F.scaled_dot_product_attention at the reference's call site
fastvideo/train_grpo_flux.py:134-144), for S % 256 == 0; the 8-wave kernel of round 1 / 2 stays for other shapes.

Same construction as the forward (csrc/gen/attn_fwd64.py: read that first): one wave per SIMD, a wave owns 64 queries =
chains a and b of 32, and every K / V / K^T fragment read from LDS feeds both chains.  Per 32-key block j and chain c:
    QK(c, j):  S^T = K Q_c^T (8 MFMAs)  and  dP^T = V dO_c^T (8 MFMAs)         [key on the row, query on the lane]
    SM(c, j):  p = exp2(s c - lse_q log2 e),  ds = p (dp - delta_q)            [4.5 VALU per element, no maximum: lse is known]
    DQ(c, j):  dQ_c^T += K^T dS_c^T (8 MFMAs), dS^T taken from the accumulator registers in place (bf16 pairs)
Block-iteration j = two segments of 24 MFMA gaps:
    segment 1:  MFMA QK(b, j), DQ(b, j-1)      VALU SM(a, j)      LDS K, V (j+1) fragments, each after its last use
    segment 2:  MFMA QK(a, j+1), DQ(a, j)      VALU SM(b, j)      LDS K^T (j) fragments
Registers: dQ_a a[0:63], dQ_b a[64:127], Q_a a[128:159], Q_b a[160:191], dO_a a[192:223], dO_b a[224:255];
           S_a v[4:19], dP_a v[20:35], S_b v[36:51], dP_b v[52:67], dS_a v[68:75], dS_b v[76:83], the block's 8 K fragments
           v[84:115], 8 V fragments v[116:147], 8 K^T fragments v[148:179].
LDS (96 KiB): two slots each of a K window, a V window (64 keys x 128, rows XOR-swizzled like the forward's K tile) and a K^T
tile ([128 d][64 keys], the forward's V^T image), by LDS-DMA.  Barrier interval t = block-iterations 2t, 2t+1: it reads K^T
tile t (keys 64 t ..) and the K / V WINDOW of keys 64 t + 32 .. 64 t + 95 (QK runs one block ahead of DQ); the window's
second half does not exist in the last interval: waves 2, 3, whose DMA pieces it is, skip them.
dQ is scaled by `scale` in the epilogue (dS carries no scale).  Checked on the CPU by tests/test_attn_bwd64_emulated.py.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from attn_fwd64 import Asm, a, ar, s, sr, v, vr  # noqa: E402

MFMA = "v_mfma_f32_32x32x16_bf16"

# ------------------------------------------------------------------------------------------------ register map
DQ_A, DQ_B = 0, 64
QF_A, QF_B = 128, 160
DOF_A, DOF_B = 192, 224
S_A, DP_A, S_B, DP_B = 4, 20, 36, 52
DS_A, DS_B = 68, 76
KFR, VFR, KTF = 84, 116, 148
KA = 180            # 8 K / V fragment read addresses (per k-step)
KTA = 188           # 4 K^T fragment read addresses (key step s = 2 kb + s2 of the tile)
KSRC = 192          # 4 K / V DMA source offsets
KTSRC = 196         # 4 K^T DMA source offsets
T = 200             # 8 temporaries
LSE_A, LSE_B, DL_A, DL_B = 208, 209, 210, 211
X = 212             # v[212:243] scratch (prologue / epilogue)
ROWOFF = 244        # (w * 64 + r) * 256 + h * 16: Q / dQ byte offset of this lane
V_LAST = 247

sW, sWOFF = 64, 65
sKP, sVP, sKTP = 66, 68, 70      # pairs: DMA base pointers
sLOOP, sROW, sKTT, sTMP = 72, 73, 74, 75
sQ, sK, sV, sKT, sDO, sLSE, sDL, sDQ = 76, 78, 80, 82, 84, 86, 88, 90
sSP2, sLDO2, sCS, sSCALE, sNLOOP, sS = 92, 93, 94, 95, 96, 97
sTP = 98                         # pair: prologue scratch pointer
sKPU, sVPU = 76, 84              # pairs (the Q / dO pointers' registers, free after the prologue): per-wave K / V DMA bases
S_FIRST, S_LAST = 64, 99

K_BASE, V_BASE, KT_BASE = 0, 32768, 65536
SLOT = 16384


class Chain:
    def __init__(self, name, DQ, QF, DOF, S, DP, DS, LSE, DL):
        self.name, self.DQ, self.QF, self.DOF, self.S, self.DP, self.DS, self.LSE, self.DL = name, DQ, QF, DOF, S, DP, DS, LSE, DL


CA = Chain("a", DQ_A, QF_A, DOF_A, S_A, DP_A, DS_A, LSE_A, DL_A)
CB = Chain("b", DQ_B, QF_B, DOF_B, S_B, DP_B, DS_B, LSE_B, DL_B)


# ------------------------------------------------------------------------------------------------ building blocks
def mfma_qk(ch, g):
    """gap g of QK: k-step ks = g >> 1; even: S^T (+)= K[ks] Q[ks]; odd: dP^T (+)= V[ks] dO[ks]."""
    ks = g >> 1
    if g & 1:
        d = vr(ch.DP, 16)
        return f"{MFMA} {d}, {vr(VFR + 4 * ks, 4)}, {ar(ch.DOF + 4 * ks, 4)}, {'0' if ks == 0 else d}"
    d = vr(ch.S, 16)
    return f"{MFMA} {d}, {vr(KFR + 4 * ks, 4)}, {ar(ch.QF + 4 * ks, 4)}, {'0' if ks == 0 else d}"


def mfma_dq(ch, n):
    """n = 4 s2 + dt: dQ^T[dt] += K^T fragment (s2, dt) x dS^T[s2]."""
    s2, dt = n >> 2, n & 3
    o = ar(ch.DQ + 16 * dt, 16)
    return f"{MFMA} {o}, {vr(KTF + 4 * n, 4)}, {vr(ch.DS + 4 * s2, 4)}, {o}"


def read_kv(g, slot, hb):
    """Reload, behind gap g of QK (its last use), the fragment that gap g of the NEXT block's QK uses: K or V rows of
    window half hb (0: rows 0..31 of the window, 1: rows 32..63)."""
    ks = g >> 1
    base = V_BASE if (g & 1) else K_BASE
    dst = VFR if (g & 1) else KFR
    return f"ds_read_b128 {vr(dst + 4 * ks, 4)}, {v(KA + ks)} offset:{base + slot * SLOT + hb * 8192}"


def read_kt(n, slot, kb):
    s2, dt = n >> 2, n & 3
    return f"ds_read_b128 {vr(KTF + 4 * n, 4)}, {v(KTA + 2 * kb + s2)} offset:{slot * SLOT + dt * 4096}"   # KT_BASE in the address


def softmax_stream(ch):
    """SM(c, j) as a flat, software-pipelined instruction list (72 instructions, 16 of them v_exp_f32)."""
    def t(e):
        return v(T + (e & 7))
    out = []
    for st in range(16 + 3):
        e = st
        if e - 1 >= 0 and e - 1 < 16:
            out.append(f"v_exp_f32 {t(e - 1)}, {t(e - 1)}")
        if e < 16:
            out.append(f"v_fma_f32 {t(e)}, {v(ch.S + e)}, {s(sCS)}, -{v(ch.LSE)}")
            out.append(f"v_sub_f32 {v(ch.DP + e)}, {v(ch.DP + e)}, {v(ch.DL)}")
        if 0 <= e - 2 < 16:
            out.append(f"v_mul_f32 {v(ch.DP + e - 2)}, {t(e - 2)}, {v(ch.DP + e - 2)}")
        if 0 <= e - 3 < 16 and ((e - 3) & 1):
            k = (e - 3) >> 1
            out.append(f"v_cvt_pk_bf16_f32 {v(ch.DS + k)}, {v(ch.DP + 2 * k)}, {v(ch.DP + 2 * k + 1)}")
    return out


def spread(items, ngaps):
    """Distribute a flat list over ngaps gaps as evenly as possible, keeping the order."""
    out, n = [], len(items)
    for g in range(ngaps):
        out.append(items[g * n // ngaps:(g + 1) * n // ngaps])
    return out


def dma_piece(kind, p, slot, ptrs=None):
    base = {"k": K_BASE, "v": V_BASE, "kt": KT_BASE}[kind] + slot * SLOT + p * 1024
    src = (KTSRC if kind == "kt" else KSRC) + p
    ptr = {"k": sKPU, "v": sVPU, "kt": sKTP}[kind] if ptrs is None else ptrs[kind]
    return (f"s_add_i32 m0, {s(sWOFF)}, {base}", f"global_load_lds_dwordx4 {v(src)}, {sr(ptr, 2)}")


def segment(A, mfmas, valu_gaps, lds=None, dma=None, waits=None):
    lds, dma, waits = lds or {}, dma or {}, waits or {}
    for g, m in enumerate(mfmas):
        if g in waits:
            A.e(waits[g])
        A.e(m)
        fill = list(valu_gaps[g]) if g < len(valu_gaps) else []
        if g in dma:
            m0w, ld = dma[g]
            A.e(m0w)
            A.e(fill.pop(0) if fill else "s_nop 0")
            A.e(ld)
        for x in fill:
            A.e(x)
        if g in lds:
            A.e(lds[g])


def block_iteration(A, par, kb, first=False, last=False, dma1=None, dma2=None):
    """Block-iteration j = 2 t + kb, t & 1 == par.  K, V (j+1) live in window half kb of slot par; K^T (j) in tile half kb.
    dma1 / dma2: DMA pieces issued inside segment 1 / segment 2."""
    A.c(f"================ block-iteration parity {par} kb {kb}{' FIRST' if first else ''}{' LAST' if last else ''}")
    # ---------------- segment 1: QK(b, j), DQ(b, j-1); SM(a, j); K, V (j+1) reloads
    mf = [mfma_qk(CB, g) for g in range(16)] + ([] if first else [mfma_dq(CB, n) for n in range(8)])
    vg = spread(softmax_stream(CA), len(mf))
    lds = {} if last else {g: read_kv(g, par, kb) for g in range(16)}
    dma = {1 + 3 * i: pc for i, pc in enumerate(dma1 or [])}
    A.c("---- segment 1")
    segment(A, mf, vg, lds=lds, dma=dma)
    if first:
        A.e("s_nop 3")                                      # no DQ gaps behind QK(b, 0): its dP is read by the next VALU
    # ---------------- segment 2: QK(a, j+1), DQ(a, j); SM(b, j); K^T (j) reloads
    mf = ([] if last else [mfma_qk(CA, g) for g in range(16)]) + [mfma_dq(CA, n) for n in range(8)]
    vg = spread(softmax_stream(CB), len(mf))
    lds = {n: read_kt(n, par, kb) for n in range(8)}
    waits = {0: "s_waitcnt lgkmcnt(0)", 16: "s_waitcnt lgkmcnt(0)"}
    if last:                                                 # K^T reads, then DQ at once: read first, wait, then the MFMAs
        for n in range(8):
            A.e(read_kt(n, par, kb))
        lds, waits = {}, {0: "s_waitcnt lgkmcnt(0)"}
    dma = {9 + 3 * i: pc for i, pc in enumerate(dma2 or [])}
    A.c("---- segment 2")
    segment(A, mf, vg, lds=lds, dma=dma, waits=waits)


def interval(A, par, first=False, last=False):
    """Barrier interval t (t & 1 == par): block-iterations (2t, 2t+1) + the DMA of interval t+1's tiles."""
    nxt = 1 - par
    if last:
        block_iteration(A, par, 0, first=first)
        block_iteration(A, par, 1, last=True)
        return
    kt = [dma_piece("kt", p, nxt) for p in range(4)]
    kp = [dma_piece("k", p, nxt) for p in range(4)]
    vp = [dma_piece("v", p, nxt) for p in range(4)]
    A.e(f"s_add_u32 {s(sKTP)}, {s(sKT)}, {s(sKTT)}")
    A.e(f"s_addc_u32 {s(sKTP + 1)}, {s(sKT + 1)}, 0")
    # rows of the next window that lie beyond the sequence (second half of the last window: waves 2, 3) are fetched from 32 rows
    # earlier instead: valid memory, and that half of the window is never read
    A.e(f"s_cmp_ge_u32 {s(sROW)}, {s(sS)}")
    A.e(f"s_cselect_b32 {s(sTMP)}, 8192, 0")
    for dst, src in ((sKPU, sKP), (sVPU, sVP)):
        A.e(f"s_sub_u32 {s(dst)}, {s(src)}, {s(sTMP)}")
        A.e(f"s_subb_u32 {s(dst + 1)}, {s(src + 1)}, 0")
    block_iteration(A, par, 0, first=first, dma1=kp, dma2=kt[:2])
    block_iteration(A, par, 1, dma1=vp, dma2=kt[2:])
    A.e(f"s_add_u32 {s(sKTT)}, {s(sKTT)}, 128")                             # next K^T tile: + 64 keys
    A.e(f"s_add_u32 {s(sROW)}, {s(sROW)}, 64")
    A.e(f"s_add_u32 {s(sKP)}, {s(sKP)}, {SLOT}")                            # next window: + 64 rows of 256 bytes
    A.e(f"s_addc_u32 {s(sKP + 1)}, {s(sKP + 1)}, 0")
    A.e(f"s_add_u32 {s(sVP)}, {s(sVP)}, {SLOT}")
    A.e(f"s_addc_u32 {s(sVP + 1)}, {s(sVP + 1)}, 0")
    A.e("s_waitcnt vmcnt(0)")
    A.e("s_barrier")


# ------------------------------------------------------------------------------------------------ prologue / epilogue
def prologue(A):
    names = (("q", sQ), ("k", sK), ("v", sV), ("kt", sKT), ("do", sDO), ("lse", sLSE), ("dl", sDL), ("dq", sDQ))
    for nm, reg in names:
        A.e(f"s_mov_b32 {s(reg)}, %[{nm}_lo]")
        A.e(f"s_mov_b32 {s(reg + 1)}, %[{nm}_hi]")
    for dst, nm in ((sSP2, "sp2"), (sLDO2, "ldo2"), (sCS, "cs"), (sSCALE, "scale"), (sNLOOP, "nloop"), (sS, "seq")):
        A.e(f"s_mov_b32 {s(dst)}, %[{nm}]")
    lane, w, r, h = v(X), v(X + 1), v(X + 2), v(X + 3)
    t0, t1, t2 = v(X + 4), v(X + 5), v(X + 6)
    A.e(f"v_and_b32 {lane}, 63, %[tid]")
    A.e(f"v_lshrrev_b32 {w}, 6, %[tid]")
    A.e(f"v_and_b32 {r}, 31, {lane}")
    A.e(f"v_lshrrev_b32 {h}, 5, {lane}")
    A.e(f"v_readfirstlane_b32 {s(sW)}, {w}")
    A.e(f"s_lshl_b32 {s(sWOFF)}, {s(sW)}, 12")
    A.c("K / V fragment read addresses (the forward's K image: MFMA row r reads window row pi(r), chunk (2 ks + h) ^ (row & 15))")
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
    A.c("K^T fragment read addresses (the forward's V^T image): row d = 32 dt + r, chunk (2 s + h) ^ ((r >> 1) & 7)")
    yv, r7 = v(X + 7), v(X + 8)
    A.e(f"v_bfe_u32 {t0}, {r}, 1, 3")
    A.e(f"v_xor_b32 {yv}, {h}, {t0}")
    A.e(f"v_lshlrev_b32 {r7}, 7, {r}")
    A.e(f"v_add_u32 {r7}, {KT_BASE}, {r7}")
    for si in range(4):
        A.e(f"v_xor_b32 {t0}, {2 * si}, {yv}")
        A.e(f"v_lshl_add_u32 {v(KTA + si)}, {t0}, 4, {r7}")
    A.c("K / V DMA source offsets: piece p of wave w = window rows 16 w + 4 p + (lane >> 4)")
    l4, l15, key0 = v(X + 7), v(X + 8), v(X + 9)
    A.e(f"v_lshrrev_b32 {l4}, 4, {lane}")
    A.e(f"v_and_b32 {l15}, 15, {lane}")
    A.e(f"v_lshl_add_u32 {key0}, {w}, 4, {l4}")
    for p in range(4):
        A.e(f"v_add_u32 {t0}, {4 * p}, {key0}")
        A.e(f"v_add_u32 {t1}, {4 * p}, {l4}")
        A.e(f"v_xor_b32 {t1}, {l15}, {t1}")
        A.e(f"v_lshlrev_b32 {t1}, 4, {t1}")
        A.e(f"v_lshl_add_u32 {v(KSRC + p)}, {t0}, 8, {t1}")
    A.c("K^T DMA source offsets: piece p of wave w = rows d = 32 w + 8 p + (lane >> 3)")
    l3, l7, d0 = v(X + 7), v(X + 8), v(X + 9)
    A.e(f"v_lshrrev_b32 {l3}, 3, {lane}")
    A.e(f"v_and_b32 {l7}, 7, {lane}")
    A.e(f"v_lshl_add_u32 {d0}, {w}, 5, {l3}")
    for p in range(4):
        A.e(f"v_add_u32 {t0}, {8 * p}, {d0}")
        A.e(f"v_bfe_u32 {t1}, {t0}, 1, 3")
        A.e(f"v_xor_b32 {t1}, {l7}, {t1}")
        A.e(f"v_lshlrev_b32 {t1}, 4, {t1}")
        A.e(f"v_mad_u32_u24 {v(KTSRC + p)}, {t0}, {s(sSP2)}, {t1}")
    A.c("this lane's query row w * 64 + r (chain b: + 32): Q / dQ offset, dO offset, lse / delta")
    row = v(X + 10)
    A.e(f"v_lshl_add_u32 {row}, {w}, 6, {r}")
    A.e(f"v_lshlrev_b32 {t1}, 8, {row}")
    A.e(f"v_lshl_add_u32 {v(ROWOFF)}, {h}, 4, {t1}")
    A.e(f"v_add_u32 {v(X + 11)}, 8192, {v(ROWOFF)}")                       # chain b Q offset
    A.e(f"v_mul_lo_u32 {t1}, {row}, {s(sLDO2)}")
    A.e(f"v_lshl_add_u32 {v(X + 12)}, {h}, 4, {t1}")                       # dO offset chain a
    A.e(f"s_lshl_b32 {s(sTMP)}, {s(sLDO2)}, 5")
    A.e(f"v_add_u32 {v(X + 13)}, {s(sTMP)}, {v(X + 12)}")                  # dO offset chain b
    A.e(f"v_lshlrev_b32 {v(X + 14)}, 2, {row}")                            # lse / delta offset
    A.c("first tiles: K^T tile 0 and the K / V window of keys 32..95 -> slot 0; keys 0..31 (the second half of 'window -1',")
    A.c("the pieces of waves 2 and 3, whose source rows are 32..63 of it) -> slot 1")
    A.e(f"s_mov_b32 {s(sKTT)}, 0")
    A.e(f"s_add_u32 {s(sKTP)}, {s(sKT)}, 0")
    A.e(f"s_addc_u32 {s(sKTP + 1)}, {s(sKT + 1)}, 0")
    for p in range(4):
        m0w, ld = dma_piece("kt", p, 0)
        A.e(m0w)
        A.e("s_nop 0")
        A.e(ld)
    A.e(f"s_mov_b32 {s(sKTT)}, 128")
    for ptr, src in ((sKP, sK), (sVP, sV)):
        A.e(f"s_add_u32 {s(ptr)}, {s(src)}, 8192")                          # window 0 starts at key 32
        A.e(f"s_addc_u32 {s(ptr + 1)}, {s(src + 1)}, 0")
    for kind in ("k", "v"):
        for p in range(4):
            m0w, ld = dma_piece(kind, p, 0, ptrs={"k": sKP, "v": sVP})
            A.e(m0w)
            A.e("s_nop 0")
            A.e(ld)
    skip = A.new_label("w01")
    A.e(f"s_cmp_lt_u32 {s(sW)}, 2")
    A.e(f"s_cbranch_scc1 {skip}")
    for ptr, src in ((sKTP, sK), (sTP, sV)):
        A.e(f"s_sub_u32 {s(ptr)}, {s(src)}, 8192")                          # window -1: rows -32..31; only its rows 32..63 are read
        A.e(f"s_subb_u32 {s(ptr + 1)}, {s(src + 1)}, 0")
    for kind in ("k", "v"):
        for p in range(4):
            m0w, ld = dma_piece(kind, p, 1, ptrs={"k": sKTP, "v": sTP})
            A.e(m0w)
            A.e("s_nop 0")
            A.e(ld)
    A.label(skip)
    # this wave's row base of the NEXT window to fetch (window 1 = keys 96..159): rows 16 w .. 16 w + 15 of it
    A.e(f"s_lshl_b32 {s(sROW)}, {s(sW)}, 4")
    A.e(f"s_add_u32 {s(sROW)}, {s(sROW)}, 96")
    A.e(f"s_add_u32 {s(sKP)}, {s(sKP)}, {SLOT}")
    A.e(f"s_addc_u32 {s(sKP + 1)}, {s(sKP + 1)}, 0")
    A.e(f"s_add_u32 {s(sVP)}, {s(sVP)}, {SLOT}")
    A.e(f"s_addc_u32 {s(sVP + 1)}, {s(sVP + 1)}, 0")
    A.c("Q and dO fragments (B operands), lse * log2(e) and delta of this lane's two rows")
    for ch, qoff, dooff in ((CA, ROWOFF, X + 12), (CB, X + 11, X + 13)):
        for ks in range(8):
            A.e(f"global_load_dwordx4 {ar(ch.QF + 4 * ks, 4)}, {v(qoff)}, {sr(sQ, 2)} offset:{32 * ks}")
        for ks in range(8):
            A.e(f"global_load_dwordx4 {ar(ch.DOF + 4 * ks, 4)}, {v(dooff)}, {sr(sDO, 2)} offset:{32 * ks}")
    A.e(f"global_load_dword {v(LSE_A)}, {v(X + 14)}, {sr(sLSE, 2)}")
    A.e(f"global_load_dword {v(LSE_B)}, {v(X + 14)}, {sr(sLSE, 2)} offset:128")
    A.e(f"global_load_dword {v(DL_A)}, {v(X + 14)}, {sr(sDL, 2)}")
    A.e(f"global_load_dword {v(DL_B)}, {v(X + 14)}, {sr(sDL, 2)} offset:128")
    A.c("dQ = 0")
    for i in range(128):
        A.e(f"v_accvgpr_write_b32 {a(i)}, 0")
    A.e("s_waitcnt vmcnt(0)")
    A.e(f"v_mul_f32 {v(LSE_A)}, 0x3fb8aa3b, {v(LSE_A)}")                   # log2(e)
    A.e(f"v_mul_f32 {v(LSE_B)}, 0x3fb8aa3b, {v(LSE_B)}")
    A.e("s_barrier")
    for g in range(16):                                                     # K, V (block 0): window half 1 of slot 1
        ks = g >> 1
        base = V_BASE if (g & 1) else K_BASE
        dst = VFR if (g & 1) else KFR
        A.e(f"ds_read_b128 {vr(dst + 4 * ks, 4)}, {v(KA + ks)} offset:{base + SLOT + 8192}")
    A.e("s_waitcnt lgkmcnt(0)")
    A.e("s_barrier")                                     # every wave holds K, V (block 0): slot 1 may be refilled
    for g in range(16):
        A.e(mfma_qk(CA, g))
    A.e("s_nop 7")
    A.e("s_nop 7")


def epilogue(A):
    A.c("================ tail: DQ(b) of the last block")
    for n in range(8):
        A.e(mfma_dq(CB, n))
    A.e("s_nop 7")
    A.e("s_nop 7")
    A.c("================ epilogue: dQ * scale -> bf16, 16-byte stores (lane halves exchanged pairwise)")
    for ch, boff in ((CA, 0), (CB, 8192)):
        for dt in range(4):
            for g in (0, 2):
                E = X + 8 + 4 * ((dt * 2 + (g >> 1)) & 3)
                rd = [v(X + 24 + j) for j in range(8)]
                for j in range(8):
                    A.e(f"v_accvgpr_read_b32 {rd[j]}, {a(ch.DQ + 16 * dt + 4 * g + j)}")
                for j in range(8):
                    A.e(f"v_mul_f32 {rd[j]}, {rd[j]}, {s(sSCALE)}")
                for j in range(4):
                    A.e(f"v_cvt_pk_bf16_f32 {v(E + j)}, {rd[2 * j]}, {rd[2 * j + 1]}")
                A.e("s_nop 1")
                A.e(f"v_permlane32_swap_b32 {v(E)}, {v(E + 2)}")
                A.e(f"v_permlane32_swap_b32 {v(E + 1)}, {v(E + 3)}")
                off = boff + 64 * dt + 16 * g
                if off < 4096:
                    A.e(f"global_store_dwordx4 {v(ROWOFF)}, {vr(E, 4)}, {sr(sDQ, 2)} offset:{off}")
                else:
                    A.e(f"global_store_dwordx4 {v(X + 7)}, {vr(E, 4)}, {sr(sDQ, 2)} offset:{off - 8192}")


def generate():
    A = Asm()
    prologue(A)
    A.e(f"v_add_u32 {v(X + 7)}, 8192, {v(ROWOFF)}")     # chain b dQ offset for the epilogue (X + 7 is free from here on)
    interval(A, 0, first=True)
    A.e(f"s_mov_b32 {s(sLOOP)}, {s(sNLOOP)}")
    loop, done = A.new_label("loop"), A.new_label("loopdone")
    A.e(f"s_cmp_eq_u32 {s(sLOOP)}, 0")
    A.e(f"s_cbranch_scc1 {done}")
    A.label(loop)
    interval(A, 1)
    interval(A, 0)
    A.e(f"s_sub_u32 {s(sLOOP)}, {s(sLOOP)}, 1")
    A.e(f"s_cmp_lg_u32 {s(sLOOP)}, 0")
    A.e(f"s_cbranch_scc1 {loop}")
    A.label(done)
    interval(A, 1, last=True)
    epilogue(A)
    return A.text()


def clobbers():
    regs = [f"v{i}" for i in range(4, V_LAST + 1)] + [f"a{i}" for i in range(256)] + \
           [f"s{i}" for i in range(S_FIRST, S_LAST + 1)] + ["vcc", "scc", "memory"]
    return ", ".join(f'"{x}"' for x in regs)


HERE = os.path.dirname(os.path.abspath(__file__))
OUT_BODY = os.path.join(HERE, "..", "attn_bwd_dq64_body.inc")


def render():
    body = generate()
    lines = ["// GENERATED by mixgrpo_amd/csrc/gen/attn_bwd_dq64.py -- do not edit; see that file for the design.",
             "#define ATTN_BWD_DQ64_CLOBBERS " + clobbers(),
             "#define ATTN_BWD_DQ64_BODY \\"]
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
