"""The generated instruction stream of attn_fwd64_kernel (mixgrpo_amd/csrc/gen/attn_fwd64.py), checked WITHOUT a GPU:
interpreted by tests/asm_emu.py (4 waves, LDS, LDS-DMA, waitcnt-delayed visibility) against an fp64 softmax(Q K^T) V of the
same bf16 operands, plus the static pass over the software-managed gfx950 hazards.  The kernel replaces
F.scaled_dot_product_attention at fastvideo/utils/sampling_utils.py:68-82 / fastvideo/train_grpo_flux.py:134-144."""
import math
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "mixgrpo_amd", "csrc", "gen"))
import asm_emu  # noqa: E402
import attn_fwd64 as G  # noqa: E402


def _bf16(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def _f32(h):
    return (h.astype(np.uint32) << 16).view(np.float32)


def _emulate(S, mode, order, seed=0, spike=False, qt=0, with_lse=True, B=1, H=1, first=None, nblk=1, stride=1, acc=False):
    """One workgroup walking `nblk` (batch, head, q-tile) blocks `stride` apart, starting at linear block `first` (default:
    q-tile `qt` of the single head).  O is [B, S, H * 128 + 64] with this tensor's columns starting at element 32.
    acc: the attn_fwd64q body on Q2 = bf16(q * scale * log2 e) (the scores are exponents of 2; the `cs` operand is not used)."""
    rng = np.random.default_rng(seed)
    nq = S // 256
    first = qt if first is None else first
    q, k, v = (rng.standard_normal((B * H, S, 128)).astype(np.float32) for _ in range(3))
    if spike:    # two keys that outscore their query's first-tile maximum by far more than 2^40: the rescale fix-up must run
        k[0, 200] = 8 * q[0, qt * 256 + 70]
        k[0, S - 3] = 6 * q[0, qt * 256 + 100]
    nt, scale = S // 64, 1 / math.sqrt(128)
    Q, K = _bf16(q * np.float32(scale * 1.4426950408889634) if acc else q), _bf16(k)
    Vt = np.ascontiguousarray(_bf16(v).transpose(0, 2, 1))
    ldo = H * 128 + 64                          # this tensor's head columns are 32 .. 32 + H * 128 (byte offset 64)
    O = np.zeros((B, S, ldo), np.uint16)
    lse = np.full((B * H, S), -7.0, np.float32)
    bh0, qt0 = first // nq, first % nq
    b0, hh0 = bh0 // H, bh0 % H
    ostep = ldo * 512
    inputs = dict(tid=np.arange(256).reshape(4, 64), q=("ptr", "Q", (bh0 * S + qt0 * 256) * 256), k=("ptr", "K", bh0 * S * 256),
                  v=("ptr", "V", bh0 * S * 256), o=("ptr", "O", (b0 * S * ldo + qt0 * 256 * ldo + hh0 * 128) * 2 + 64),
                  l=("ptr", "L", (bh0 * S + qt0 * 256) * 4), sp2=S * 2, ldo2=ldo * 2,
                  cs=(float("nan") if acc else float(np.float32(scale * 1.4426950408889634))), nloop=(nt - 2) // 2, kmax=(nt - 1) * 16384,
                  vmax=(nt - 1) * 128, nblk=nblk, qt0=qt0, hh0=hh0, b0=b0, nq=nq, nh=H, kstep=S * 256, ostep=ostep,
                  obs=S * ldo * 2, ob=("ptr", "O", 64), sq=stride % nq, dbh=stride // nq, qstride=stride * 65536,
                  lstride=stride * 1024)
    if not with_lse:
        inputs = {**{k_: v_ for k_, v_ in inputs.items() if k_ != "l"}, "l_lo": 0, "l_hi": 0}
    m = asm_emu.Machine(G.generate(acc=acc), inputs, dict(Q=Q, K=K, V=Vt, O=O, L=lse), mode=mode, order=order).run()
    rels, lerrs = [], []
    touched = np.zeros_like(O, dtype=bool)
    for blk in range(first, first + nblk * stride, stride):
        bh, qb = blk // nq, blk % nq
        b, hh = bh // H, bh % H
        qf, kf, vf = (_f32(x[bh]).astype(np.float64) for x in (Q, K, _bf16(v)))
        s = qf[qb * 256:qb * 256 + 256] @ kf.T * (math.log(2.0) if acc else scale)
        mx = s.max(1, keepdims=True)
        p = np.exp(s - mx)
        ref = p @ vf / p.sum(1, keepdims=True)
        rows, cols = slice(qb * 256, qb * 256 + 256), slice(32 + hh * 128, 160 + hh * 128)
        got = _f32(O[b, rows, cols]).astype(np.float64)
        touched[b, rows, cols] = True
        rels.append(np.linalg.norm(got - ref) / np.linalg.norm(ref))
        if with_lse:
            lerrs.append(np.abs(lse[bh, rows] - (mx[:, 0] + np.log(p.sum(1)))).max())
            lse[bh, rows] = -7.0
    assert not O[~touched].any(), "stores outside the workgroup's O blocks"
    assert (lse == -7.0).all(), "lse stores outside the workgroup's blocks"
    return max(rels), (max(lerrs) if lerrs else 0.0), lse, m


ACC = pytest.mark.parametrize("acc", [False, True], ids=["fwd64", "fwd64q"])


@ACC
def test_generated_file_is_current(acc):
    with open(G.OUT_BODY_Q if acc else G.OUT_BODY) as f:
        assert f.read() == G.render(acc=acc), "run `python mixgrpo_amd/csrc/gen/attn_fwd64.py` (or mixgrpo_amd.build)"


def test_q_variant_has_no_scale_arithmetic_in_the_loop():
    """attn_fwd64q: no v_fma_f32 anywhere, the scale operand only copied in, one subtract per score in the FIRST tile of a
    block only (64 per block against 64 fused multiply-adds per TILE in attn_fwd64), and every later score tile starts its
    accumulator from the chain's -m block."""
    t, tq = G.generate(), G.generate(acc=True)
    assert t.count("v_fma_f32") >= 4 * 64 and tq.count("v_fma_f32") == 0
    assert sum(1 for ln in tq.split("\n") if "s86" in ln) == 1
    lines = tq.split("\n")
    lo = next(i for i, ln in enumerate(lines) if ln.startswith(".Lloop_"))
    hi = next(i for i, ln in enumerate(lines) if ln.startswith(".Lloopdone_"))
    loop = lines[lo:hi]
    assert not any("v_sub_f32" in ln for ln in loop)
    starts = [ln for ln in loop if "v_mfma" in ln and ln.rstrip().endswith(("v[240:255]", "v[216:231]"))]
    assert len(starts) == 2 * 2 * 2            # two iterations x two chains x two key halves
    assert not any(ln.rstrip().endswith(", 0") for ln in loop if "v_mfma" in ln)


@ACC
def test_static_hazards_clean(acc):
    text = G.generate(acc=acc)
    assert asm_emu.check_hazards(text) == []
    # the loop body once more behind itself: hazards across the back edge
    lines = text.split("\n")
    lo = next(i for i, ln in enumerate(lines) if ln.startswith(".Lloop_"))
    hi = next(i for i, ln in enumerate(lines) if ln.startswith(".Lloopdone_"))
    assert asm_emu.check_hazards("\n".join(lines[lo + 1:hi] + lines[lo + 1:hi])) == []
    assert text.count("v_mfma_f32_32x32x16_bf16") == 256


def test_hazard_checker_sees_a_planted_hazard():
    bad = "v_mfma_f32_32x32x16_bf16 v[4:19], a[0:3], a[4:7], 0\nv_add_f32 v40, v4, v5\n"
    assert any(x.startswith("R1") for x in asm_emu.check_hazards(bad))
    bad = "v_exp_f32 v1, v2\nv_add_f32 v3, v1, v1\n"
    assert any(x.startswith("R3") for x in asm_emu.check_hazards(bad))
    bad = "v_cvt_pk_bf16_f32 v68, v1, v2\nv_mfma_f32_32x32x16_bf16 a[0:15], v[100:103], v[68:71], a[0:15]\n"
    assert any(x.startswith("R2") for x in asm_emu.check_hazards(bad))


@ACC
@pytest.mark.parametrize("mode,order", [("late", [0, 1, 2, 3]), ("early", [3, 2, 1, 0]), ("early", [0, 1, 2, 3])])
def test_emulated_vs_reference_two_tiles_per_loop(mode, order, acc):
    rel, lse_err, _, m = _emulate(256, mode, order, acc=acc)              # 4 tiles: first, one loop trip, last
    assert rel < 4e-3 and lse_err < 1e-5
    assert m.mfma_count == 4 * 4 * 64
    assert not any(k.startswith(".Lfix") for k in m.branches_taken)


@ACC
def test_emulated_second_query_block_and_longer_loop(acc):
    rel, lse_err, _, _ = _emulate(512, "late", [2, 0, 3, 1], seed=1, qt=1, acc=acc)
    assert rel < 4e-3 and lse_err < 1e-5


@ACC
@pytest.mark.parametrize("mode", ["late", "early"])
def test_emulated_rescale_fixup_runs_and_is_right(mode, acc):
    rel, lse_err, _, m = _emulate(512, mode, [1, 3, 0, 2], seed=2, spike=True, acc=acc)
    assert any(k.startswith(".Lfix") for k in m.branches_taken), "the spiked keys did not force the rescale path"
    assert rel < 4e-3 and lse_err < 1e-4


@ACC
def test_emulated_null_lse_pointer_stores_nothing(acc):
    rel, _, lse, _ = _emulate(256, "late", [0, 1, 2, 3], seed=3, with_lse=False, H=2, nblk=2, acc=acc)
    assert rel < 4e-3 and (lse == -7.0).all()


@ACC
@pytest.mark.parametrize("mode,order", [("late", [0, 1, 2, 3]), ("early", [3, 1, 2, 0])])
def test_emulated_persistent_blocks_cross_heads_and_batches(mode, order, acc):
    """One workgroup walking five consecutive blocks of a [B = 2, H = 2, S = 512] problem, starting at the second q-tile of
    (batch 0, head 0): q-tile wrap -> next head (K / V^T advance, O moves 128 columns), head wrap -> next batch; the next block's
    tiles and Q fragments are fetched during the current block's last iteration behind a counted vmcnt."""
    rel, lse_err, _, m = _emulate(512, mode, order, seed=5, B=2, H=2, first=1, nblk=5, acc=acc)
    assert rel < 4e-3 and lse_err < 1e-5
    assert m.mfma_count == 5 * 4 * 8 * 64


def test_emulated_persistent_blocks_strided():
    """Blocks 1, 4, 7, 10 of a [B = 2, H = 3, S = 512] problem (stride 3 > nq = 2: the head index advances by one or two per
    step and wraps into the next batch), the order the XCD-interleaved launch walks."""
    rel, lse_err, _, _ = _emulate(512, "late", [0, 1, 2, 3], seed=6, B=2, H=3, first=1, nblk=4, stride=3)
    assert rel < 4e-3 and lse_err < 1e-5


def test_emulated_walk_needs_stride_over_nq_below_heads():
    """The walk carries the head index ONCE per step, so a stride that moves it by H or more (stride / nq >= H) stores outside
    the workgroup's O blocks: `mgx_attn_fwd` must not dispatch such shapes to this kernel (its `walk_ok` guard; round 3's
    advisor case S = 256, B = 3, H = 2, stride 5).  The control with stride / nq < H passes."""
    with pytest.raises(AssertionError, match="stores outside"):
        _emulate(256, "late", [0, 1, 2, 3], B=3, H=2, first=0, nblk=2, stride=5)
    rel, lse_err, _, _ = _emulate(256, "late", [0, 1, 2, 3], B=3, H=2, first=0, nblk=2, stride=1)
    assert rel < 4e-3 and lse_err < 1e-5


def test_dispatch_guard_matches_the_walk():
    """The host-side guard of csrc/attention.hip restated: grid = min(256, blocks), stride = grid / 8 (or grid), and the
    64-wide kernel is taken only when stride / nq < H.  FLUX shapes pass; the advisor's failing shapes do not."""
    def walk_ok(S, H, B):
        nq = S // 256
        nblk = nq * H * B
        grid = min(256, nblk)
        stride = grid // 8 if grid % 8 == 0 else grid
        return stride // nq < H
    assert walk_ok(4608, 24, 8) and walk_ok(4608, 24, 1) and walk_ok(1536, 24, 4) and walk_ok(768, 24, 7)
    assert not walk_ok(256, 24, 11)            # stride 32, nq 1: 32 >= 24
    assert not walk_ok(512, 16, 9)             # stride 32, nq 2: 16 >= 16
    import re, os
    src = open(os.path.join(os.path.dirname(__file__), "..", "mixgrpo_amd", "csrc", "attention.hip")).read()
    assert re.search(r"walk_ok = S >= 256 && stride64 / \(S / 256\) < H", src)
