"""The generated instruction streams of the 64-wide attention BACKWARD kernels (mixgrpo_amd/csrc/gen/attn_bwd_dq64.py, ...),
checked without a GPU: interpreted by tests/asm_emu.py against an fp64 reference of the same bf16 operands (dQ = scale *
dS K, dS = P o (dP - delta), P = exp(scale Q K^T - lse)) and passed through the static gfx950 hazard check.  The kernels
replace the autograd of F.scaled_dot_product_attention at fastvideo/train_grpo_flux.py:134-144."""
import math
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "mixgrpo_amd", "csrc", "gen"))
import asm_emu  # noqa: E402
import attn_bwd_dkv64 as GK  # noqa: E402
import attn_bwd_dq64 as GQ  # noqa: E402


def _bf16(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def _f32(h):
    return (h.astype(np.uint32) << 16).view(np.float32)


def _problem(S, seed, ldo=256, col0=32):
    rng = np.random.default_rng(seed)
    q, k, v = (rng.standard_normal((S, 128)).astype(np.float32) for _ in range(3))
    do = rng.standard_normal((S, ldo)).astype(np.float32)
    Q, K, V, DO = _bf16(q), _bf16(k), _bf16(v), _bf16(do)
    scale = 1 / math.sqrt(128)
    qf, kf, vf, dof = (_f32(x).astype(np.float64) for x in (Q, K, V, DO[:, col0:col0 + 128]))
    s = qf @ kf.T * scale
    lse = np.log(np.exp(s - s.max(1, keepdims=True)).sum(1)) + s.max(1)
    P = np.exp(s - lse[:, None])
    delta = (dof * (P @ vf)).sum(1)
    dS = P * (dof @ vf.T - delta[:, None])
    ref = dict(dQ=dS @ kf * scale, dK=dS.T @ qf * scale, dV=P.T @ dof)
    return dict(Q=Q, K=K, V=V, DO=DO, KT=np.ascontiguousarray(K.T), QT=np.ascontiguousarray(Q.T), LSE=lse.astype(np.float32),
                DL=delta.astype(np.float32), scale=scale, ldo=ldo, col0=col0), ref


def _emulate_dq(S, mode, order, seed=0, qt=0):
    pr, ref = _problem(S, seed)
    dQ = np.zeros((S, 128), np.uint16)
    nt, ldo = S // 64, pr["ldo"]
    inputs = dict(tid=np.arange(256).reshape(4, 64), q=("ptr", "Q", qt * 65536), k=("ptr", "K", 0), v=("ptr", "V", 0),
                  kt=("ptr", "KT", 0), do=("ptr", "DO", qt * 256 * ldo * 2 + 2 * pr["col0"]), lse=("ptr", "LSE", qt * 1024),
                  dl=("ptr", "DL", qt * 1024), dq=("ptr", "DQ", qt * 65536), sp2=S * 2, ldo2=ldo * 2,
                  cs=float(np.float32(pr["scale"] * 1.4426950408889634)), scale=float(np.float32(pr["scale"])),
                  nloop=(nt - 2) // 2, seq=S)
    bufs = {k_: pr[k_] for k_ in ("Q", "K", "V", "KT", "DO", "LSE", "DL")}
    bufs["DQ"] = dQ
    m = asm_emu.Machine(GQ.generate(), inputs, bufs, lds_bytes=98304, mode=mode, order=order).run()
    got = _f32(dQ[qt * 256:qt * 256 + 256]).astype(np.float64)
    want = ref["dQ"][qt * 256:qt * 256 + 256]
    assert not np.delete(dQ, np.s_[qt * 256:qt * 256 + 256], axis=0).any(), "stores outside this workgroup's dQ block"
    return np.linalg.norm(got - want) / np.linalg.norm(want), m


def _emulate_dkv(S, mode, order, seed=0, kt=0):
    pr, ref = _problem(S, seed)
    dK, dV = np.zeros((S, 128), np.uint16), np.zeros((S, 128), np.uint16)
    DOT = np.ascontiguousarray(pr["DO"][:, pr["col0"]:pr["col0"] + 128].T)
    NQ, ldo = S // 32, pr["ldo"]
    inputs = dict(tid=np.arange(256).reshape(4, 64), q=("ptr", "Q", 0), do=("ptr", "DO", 2 * pr["col0"]), qt=("ptr", "QT", 0),
                  dot=("ptr", "DOT", 0), k=("ptr", "K", kt * 65536), v=("ptr", "V", kt * 65536), lse=("ptr", "LSE", 0),
                  dl=("ptr", "DL", 0), dk=("ptr", "DK", kt * 65536), dv=("ptr", "DV", kt * 65536), sp2=S * 2, ldo2=ldo * 2,
                  cs=float(np.float32(pr["scale"] * 1.4426950408889634)), scale=float(np.float32(pr["scale"])),
                  nis=float(np.float32(-1.0 / pr["scale"])), nloop=(NQ - 2) // 2, qmax=(NQ - 1) * 8192, ldo32=64 * ldo,
                  cmax=(NQ - 1) * 128)
    bufs = {k_: pr[k_] for k_ in ("Q", "K", "V", "QT", "DO", "LSE", "DL")}
    bufs.update(DOT=DOT, DK=dK, DV=dV)
    m = asm_emu.Machine(GK.generate(), inputs, bufs, lds_bytes=GK.LDS_BYTES, mode=mode, order=order).run()
    sl = slice(kt * 256, kt * 256 + 256)
    rels = []
    for nm, arr in (("dK", dK), ("dV", dV)):
        got, want = _f32(arr[sl]).astype(np.float64), ref[nm][sl]
        rels.append(np.linalg.norm(got - want) / np.linalg.norm(want))
        assert not np.delete(arr, np.s_[sl], axis=0).any(), "stores outside this workgroup's key block"
    return rels, m


def test_generated_files_are_current():
    for G in (GQ, GK):
        with open(G.OUT_BODY) as f:
            assert f.read() == G.render(), "run `python -m mixgrpo_amd.build`"


def test_dkv64_static_hazards_clean():
    text = GK.generate()
    assert asm_emu.check_hazards(text) == []
    lines = text.split("\n")
    lo = next(i for i, ln in enumerate(lines) if ln.startswith(".Lloop_"))
    hi = next(i for i, ln in enumerate(lines) if ln.startswith(".Lloopdone_"))
    assert asm_emu.check_hazards("\n".join(lines[lo + 1:hi] + lines[lo + 1:hi])) == []


@pytest.mark.parametrize("mode,order", [("late", [0, 1, 2, 3]), ("early", [3, 2, 1, 0])])
def test_dkv64_emulated_vs_reference(mode, order):
    (rk, rv), m = _emulate_dkv(256, mode, order)            # 8 query blocks: first, three loop trips, last, tail
    assert rk < 4e-3 and rv < 4e-3
    assert m.mfma_count == 4 * 8 * 64


@pytest.mark.parametrize("mode,order", [("late", [2, 0, 3, 1]), ("early", [1, 3, 0, 2])])
def test_dkv64_emulated_second_key_block(mode, order):
    (rk, rv), _ = _emulate_dkv(512, mode, order, seed=1, kt=1)
    assert rk < 4e-3 and rv < 4e-3


def test_dq64_static_hazards_clean():
    text = GQ.generate()
    assert asm_emu.check_hazards(text) == []
    lines = text.split("\n")
    lo = next(i for i, ln in enumerate(lines) if ln.startswith(".Lloop_"))
    hi = next(i for i, ln in enumerate(lines) if ln.startswith(".Lloopdone_"))
    assert asm_emu.check_hazards("\n".join(lines[lo + 1:hi] + lines[lo + 1:hi])) == []


@pytest.mark.parametrize("mode,order", [("late", [0, 1, 2, 3]), ("early", [3, 2, 1, 0])])
def test_dq64_emulated_vs_reference(mode, order):
    rel, m = _emulate_dq(256, mode, order)                  # 4 key tiles: first interval, one loop trip, last
    assert rel < 4e-3
    assert m.mfma_count == 4 * 8 * 48


@pytest.mark.parametrize("mode,order", [("late", [2, 0, 3, 1]), ("early", [1, 3, 0, 2])])
def test_dq64_emulated_second_query_block(mode, order):
    rel, _ = _emulate_dq(512, mode, order, seed=1, qt=1)
    assert rel < 4e-3
