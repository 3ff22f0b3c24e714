#!/usr/bin/env python3
"""SCRATCH PROTOTYPE (not product): K-loop of a bf16 GEMM at ONE wave per SIMD, as a hand-placed gfx950 instruction stream --
the measurement behind DESIGN.md's "GEMM as a generated instruction stream" lever.

C[M, N] (bf16) = A[M, K] W[N, K]^T, M, N % 256 == 0, K % 128 == 0, plain row-major operands (lda = ldw = K, ldc = N); one
workgroup (4 waves, 512 registers each) per 256 x 256 output tile, wave (wm, wn) owns 128 x 128 of it as 4 x 4 blocks of
v_mfma_f32_32x32x16_bf16 (MFMA A operand = W fragment, rows = features; B operand = A fragment, columns = tokens; the W rows of a
32-block are permuted through the fragment read address so that a lane's 16 accumulator registers are 16 consecutive features).
K-tiles of 64 by LDS-DMA into two stages of (A 32 KiB | W 32 KiB) = 128 KiB, XOR-swizzled through the per-lane source address;
a K-tile = 4 k-steps of 16 MFMAs; the fragments of k-step s live in register buffer s (4 x 32 registers) and buffer s is refilled
during k-step s + 1 (one ds_read_b128 per MFMA gap, 8 per k-step: HALF the LDS reads per MFMA of the 8-wave kernel); the 16 DMA
pieces of K-tile j + 2 are issued in k-steps 1 and 2 of K-tile j; ONE barrier per K-tile (after k-step 0) serves both the stage
hand-over and the DMA's visibility.  The epilogue is NOT the lever measured here (plain bf16 stores after the loop).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
MFMA = "v_mfma_f32_32x32x16_bf16"

# vector registers
WA, AA = 4, 8             # 4 + 4 fragment read addresses (per k-step)
DS = 12                   # 8 DMA source byte offsets (pieces q = 0..7; A and W share them: lda == ldw)
T = 20                    # temporaries v[20:43]
CO = 44                   # 4 output offsets (per token block j)
FB = 64                   # fragment buffers v[64:191]: buffer s = v[64 + 32 s : +32] = W frags i (4 regs each) | A frags j
EP = 192                  # epilogue scratch v[192:223]
V_FIRST, V_LAST = 4, 223
# scalar registers
sA, sW, sC = 64, 66, 68
sLDA2, sLDC2, sNLOOP, sKMAX, sKOFF = 70, 71, 72, 73, 74
sAP, sWP = 76, 78
sWID, sWOFF, sTMP, sTMP2, sWM, sWN = 80, 81, 82, 83, 84, 85
S_FIRST, S_LAST = 64, 85
W_LDS = 65536
STAGE = 32768


def v(i): return f"v{i}"
def vr(i, n): return f"v[{i}:{i + n - 1}]"
def a(i): return f"a{i}"
def ar(i, n): return f"a[{i}:{i + n - 1}]"
def s(i): return f"s{i}"
def sr(i, n): return f"s[{i}:{i + n - 1}]"


class Asm:
    def __init__(self):
        self.lines = []

    def e(self, t):
        self.lines.append("  " + t)

    def c(self, t):
        self.lines.append("  ; " + t)

    def label(self, n):
        self.lines.append(f"{n}:")

    def text(self):
        return "\n".join(self.lines) + "\n"


def wfrag(buf, i): return vr(FB + 32 * buf + 4 * i, 4)
def afrag(buf, j): return vr(FB + 32 * buf + 16 + 4 * j, 4)
def acc(i, j): return ar((4 * i + j) * 16, 16)


def reads(buf, ks, stage):
    """the 8 fragment reads of k-step ks of the K-tile in `stage` into buffer `buf`"""
    out = [f"ds_read_b128 {wfrag(buf, i)}, {v(WA + ks)} offset:{stage * STAGE + 4096 * i}" for i in range(4)]
    out += [f"ds_read_b128 {afrag(buf, j)}, {v(AA + ks)} offset:{stage * STAGE + 4096 * j}" for j in range(4)]
    return out


def dma(kind, q, stage):
    base = (0 if kind == "a" else W_LDS) + stage * STAGE + q * 1024
    ptr = sAP if kind == "a" else sWP
    return [f"s_add_u32 m0, {s(sWOFF)}, {base}", "s_nop 0", f"global_load_lds_dwordx4 {v(DS + q)}, {sr(ptr, 2)}"]


def kstep(A, buf, first_tile_zero=False, per_gap=None):
    """16 MFMAs of buffer `buf`; per_gap[g] = instructions placed behind MFMA g"""
    per_gap = per_gap or {}
    g = 0
    for i in range(4):
        for j in range(4):
            c_in = "0" if first_tile_zero else acc(i, j)
            A.e(f"{MFMA} {acc(i, j)}, {wfrag(buf, i)}, {afrag(buf, j)}, {c_in}")
            for x in per_gap.get(g, []):
                A.e(x)
            g += 1


def ktile(A, P, first=False):
    """K-tile in stage P; the next one in stage 1 - P"""
    A.c(f"---------------- K-tile, stage {P}{' (first of the tile: accumulators start at 0)' if first else ''}")
    A.e("s_waitcnt lgkmcnt(15)")                       # buffer 0 (read three k-steps ago) is complete
    r = reads(3, 3, P)
    kstep(A, 0, first_tile_zero=first, per_gap={g: [r[g]] for g in range(8)})
    A.e("s_waitcnt vmcnt(0) lgkmcnt(0)")               # own DMA pieces of the next K-tile; own reads of this stage
    A.e("s_barrier")                                   # stage P is free, stage 1 - P holds the next K-tile for everybody
    r = reads(0, 0, 1 - P)
    gaps = {g: [r[g]] for g in range(8)}
    gaps[0] = [f"s_add_u32 {s(sAP)}, {s(sA)}, {s(sKOFF)}", f"s_addc_u32 {s(sAP + 1)}, {s(sA + 1)}, 0"] + gaps[0]
    gaps[1] = [f"s_add_u32 {s(sWP)}, {s(sW)}, {s(sKOFF)}", f"s_addc_u32 {s(sWP + 1)}, {s(sW + 1)}, 0"] + gaps[1]
    for q in range(8):
        gaps[8 + q] = dma("a", q, P)
    kstep(A, 1, per_gap=gaps)
    r = reads(1, 1, 1 - P)
    gaps = {g: [r[g]] for g in range(8)}
    for q in range(8):
        gaps[8 + q] = dma("w", q, P)
    kstep(A, 2, per_gap=gaps)
    r = reads(2, 2, 1 - P)
    gaps = {g: [r[g]] for g in range(8)}
    gaps[8] = [f"s_add_u32 {s(sKOFF)}, {s(sKOFF)}, 128", f"s_min_u32 {s(sKOFF)}, {s(sKOFF)}, {s(sKMAX)}"]
    kstep(A, 3, per_gap=gaps)


def generate():
    A = Asm()
    for dst, name in ((sA, "a_lo"), (sA + 1, "a_hi"), (sW, "w_lo"), (sW + 1, "w_hi"), (sC, "c_lo"), (sC + 1, "c_hi"),
                      (sLDA2, "lda2"), (sLDC2, "ldc2"), (sNLOOP, "nloop"), (sKMAX, "kmax")):
        A.e(f"s_mov_b32 {s(dst)}, %[{name}]")
    lane, w, r, h = v(T), v(T + 1), v(T + 2), v(T + 3)
    t0, t1, t2, pr = v(T + 4), v(T + 5), v(T + 6), v(T + 7)
    A.e(f"v_and_b32 {lane}, 63, %[tid]")
    A.e(f"v_lshrrev_b32 {w}, 6, %[tid]")
    A.e(f"v_and_b32 {r}, 31, {lane}")
    A.e(f"v_lshrrev_b32 {h}, 5, {lane}")
    A.e(f"v_readfirstlane_b32 {s(sWID)}, {w}")
    A.e(f"s_lshl_b32 {s(sWOFF)}, {s(sWID)}, 13")                 # this wave's 8 pieces of each operand tile: 8 KiB
    A.e(f"s_lshr_b32 {s(sWM)}, {s(sWID)}, 1")
    A.e(f"s_and_b32 {s(sWN)}, {s(sWID)}, 1")
    A.c("W fragment rows: MFMA row r reads tile row pi(r) = 16 ((r >> 2) & 1) + 4 ((r >> 3) & 3) + (r & 3)")
    A.e(f"v_and_b32 {t0}, 3, {r}")
    A.e(f"v_bfe_u32 {t1}, {r}, 3, 2")
    A.e(f"v_lshl_or_b32 {t0}, {t1}, 2, {t0}")
    A.e(f"v_bfe_u32 {t1}, {r}, 2, 1")
    A.e(f"v_lshl_or_b32 {pr}, {t1}, 4, {t0}")
    A.c("fragment read addresses: row * 128 + ((2 ks + h) ^ ((row >> 1) & 7)) * 16")
    for (dst, rowreg, base_s, lds0) in ((WA, pr, sWN, W_LDS), (AA, r, sWM, 0)):
        A.e(f"v_bfe_u32 {t1}, {rowreg}, 1, 3")                   # (row >> 1) & 7
        A.e(f"s_lshl_b32 {s(sTMP)}, {s(base_s)}, 14")            # 128 rows x 128 bytes
        if lds0:
            A.e(f"s_add_u32 {s(sTMP)}, {s(sTMP)}, {lds0}")
        A.e(f"v_lshl_add_u32 {t2}, {rowreg}, 7, {s(sTMP)}")
        for ks in range(4):
            A.e(f"v_or_b32 {t0}, {2 * ks}, {h}")                 # 2 ks + h (h is 0 / 1)
            A.e(f"v_xor_b32 {t0}, {t0}, {t1}")
            A.e(f"v_lshl_add_u32 {v(dst + ks)}, {t0}, 4, {t2}")
    A.c("DMA source offsets: piece q = rows 64 w + 8 q + (lane >> 3), chunk (lane & 7) ^ ((4 q + (lane >> 4)) & 7)")
    A.e(f"v_lshrrev_b32 {t0}, 3, {lane}")                        # row in piece
    A.e(f"v_and_b32 {t1}, 7, {lane}")                            # slot
    A.e(f"v_lshrrev_b32 {t2}, 4, {lane}")
    A.e(f"s_lshl_b32 {s(sTMP)}, {s(sWID)}, 6")
    for q in range(8):
        A.e(f"v_add_u32 {v(T + 8)}, {4 * (q & 1)}, {t2}")
        A.e(f"v_xor_b32 {v(T + 8)}, {v(T + 8)}, {t1}")           # chunk
        A.e(f"s_add_u32 {s(sTMP2)}, {s(sTMP)}, {8 * q}")
        A.e(f"v_add_u32 {v(T + 9)}, {s(sTMP2)}, {t0}")           # tile row
        A.e(f"v_mul_lo_u32 {v(T + 9)}, {v(T + 9)}, {s(sLDA2)}")
        A.e(f"v_lshl_add_u32 {v(DS + q)}, {v(T + 8)}, 4, {v(T + 9)}")
    A.c("output offsets: token (wm 128 + 32 j + r) * ldc2 + (wn 128 + 16 h) * 2")
    A.e(f"s_lshl_b32 {s(sTMP)}, {s(sWM)}, 7")
    A.e(f"v_add_u32 {t0}, {s(sTMP)}, {r}")
    A.e(f"s_lshl_b32 {s(sTMP)}, {s(sWN)}, 8")
    A.e(f"v_lshl_add_u32 {t1}, {h}, 5, {s(sTMP)}")
    for j in range(4):
        A.e(f"v_add_u32 {t2}, {32 * j}, {t0}")
        A.e(f"v_mul_lo_u32 {t2}, {t2}, {s(sLDC2)}")
        A.e(f"v_add_u32 {v(CO + j)}, {t2}, {t1}")
    A.c("prologue: K-tiles 0, 1 -> stages 0, 1")
    A.e(f"s_mov_b32 {s(sAP)}, {s(sA)}"); A.e(f"s_mov_b32 {s(sAP + 1)}, {s(sA + 1)}")
    A.e(f"s_mov_b32 {s(sWP)}, {s(sW)}"); A.e(f"s_mov_b32 {s(sWP + 1)}, {s(sW + 1)}")
    for st in range(2):
        for q in range(8):
            for x in dma("a", q, st):
                A.e(x)
        for q in range(8):
            for x in dma("w", q, st):
                A.e(x)
        if st == 0:
            A.e(f"s_min_u32 {s(sKOFF)}, 128, {s(sKMAX)}")
            A.e(f"s_add_u32 {s(sAP)}, {s(sA)}, {s(sKOFF)}"); A.e(f"s_addc_u32 {s(sAP + 1)}, {s(sA + 1)}, 0")
            A.e(f"s_add_u32 {s(sWP)}, {s(sW)}, {s(sKOFF)}"); A.e(f"s_addc_u32 {s(sWP + 1)}, {s(sW + 1)}, 0")
    A.e(f"s_min_u32 {s(sKOFF)}, 256, {s(sKMAX)}")
    A.e("s_waitcnt vmcnt(0)")
    A.e("s_barrier")
    for ks in range(3):
        for x in reads(ks, ks, 0):
            A.e(x)
    A.e("s_waitcnt lgkmcnt(0)")
    loop, done = ".Lkloop_%=", ".Lkdone_%="
    ktile(A, 0, first=True)
    ktile(A, 1)
    A.e(f"s_cmp_eq_u32 {s(sNLOOP)}, 1")
    A.e(f"s_cbranch_scc1 {done}")
    A.e(f"s_sub_u32 {s(sNLOOP)}, {s(sNLOOP)}, 1")
    A.label(loop)
    ktile(A, 0)
    ktile(A, 1)
    A.e(f"s_sub_u32 {s(sNLOOP)}, {s(sNLOOP)}, 1")
    A.e(f"s_cmp_lg_u32 {s(sNLOOP)}, 0")
    A.e(f"s_cbranch_scc1 {loop}")
    A.label(done)
    A.e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    A.c("epilogue: lane (token, h) holds features 16 h .. 16 h + 15 of every block")
    A.e("s_nop 7"); A.e("s_nop 7")
    for i in range(4):
        for j in range(4):
            for e in range(2):
                for k in range(8):
                    A.e(f"v_accvgpr_read_b32 {v(EP + k)}, {a((4 * i + j) * 16 + 8 * e + k)}")
                for k in range(4):
                    A.e(f"v_cvt_pk_bf16_f32 {v(EP + 8 + k)}, {v(EP + 2 * k)}, {v(EP + 2 * k + 1)}")
                A.e("s_nop 1")
                A.e(f"global_store_dwordx4 {v(CO + j)}, {vr(EP + 8, 4)}, {sr(sC, 2)} offset:{64 * i + 16 * e}")
    A.e("s_waitcnt vmcnt(0)")
    return A.text()


def clobbers():
    regs = [f"v{i}" for i in range(V_FIRST, V_LAST + 1)] + [f"a{i}" for i in range(256)] + [f"s{i}" for i in range(S_FIRST, S_LAST + 1)]
    return ", ".join(f'"{r}"' for r in regs + ["scc", "vcc", "memory"])


def render():
    body = generate()
    lines = ["// GENERATED by scratch/gemm4w/gen.py (prototype)", "#define GEMM4W_CLOBBERS " + clobbers(), "#define GEMM4W_BODY \\"]
    for ln in body.rstrip("\n").split("\n"):
        lines.append('  "' + ln.replace("\\", "\\\\").replace('"', '\\"') + '\\n" \\')
    lines.append('  ""')
    return "\n".join(lines) + "\n"


if __name__ == "__main__":
    if "--print" in sys.argv:
        sys.stdout.write(generate())
    else:
        with open(os.path.join(HERE, "gemm4w_body.inc"), "w") as f:
            f.write(render())
        print("written")
