"""The prescaled-Q attention path on the GPU: `mgx_qk_norm_rope_fwd_qs / _bwd_qs` (Q leaves as bf16(q * scale * log2 e)),
`mgx_attn_fwd_log2` (64-query kernel attn_fwd64q: the score tile's MFMA accumulator starts at -m, csrc/gen/attn_fwd64.py ACC;
other shapes: the 8-wave kernel with a unit exponent scale) and `mgx_attn_bwd` at scale = ln 2 -- the F.scaled_dot_product_attention
call sites of fastvideo/utils/sampling_utils.py:68-82 / fastvideo/train_grpo_flux.py:134-144.  References are fp32 torch on the
same bf16 operands (a floating-point kernel: tolerances written at each assertion)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

C = 1.4426950408889634 / math.sqrt(128)
LN2 = math.log(2.0)


def rel_err(a, b):
    a, b = a.detach().float(), b.detach().float()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def _qkv_case(B, H, S, rows, s0, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    qkv = torch.randn(B * rows, 3 * H * 128, device="cuda", generator=g).bfloat16()
    wq = 1 + 0.1 * torch.randn(128, device="cuda", generator=g)
    wk = 1 + 0.1 * torch.randn(128, device="cuda", generator=g)
    ang = torch.rand(S, 64, device="cuda", generator=g) * 6.28
    cos = torch.cos(ang).repeat_interleave(2, dim=1).contiguous()
    sin = torch.sin(ang).repeat_interleave(2, dim=1).contiguous()
    return qkv, wq, wk, cos, sin


def _run_fwd(ops, qkv, wq, wk, cos, sin, B, H, S, Sp, rows, s0, q_scale, extras):
    Q = torch.zeros(B, H, S, 128, device="cuda", dtype=torch.bfloat16)
    K = torch.zeros_like(Q)
    Vt = torch.zeros(B, H, 128, Sp, device="cuda", dtype=torch.bfloat16)
    ex = {}
    if extras:
        ex = dict(V=torch.zeros_like(Q), Qt=torch.zeros_like(Vt), Kt=torch.zeros_like(Vt))
    ops.qk_norm_rope(qkv, wq, wk, cos, sin, Q, K, Vt, B, H, S, Sp, rows, s0, q_scale=q_scale, **ex)
    return Q, K, Vt, ex


@pytest.mark.parametrize("extras", [False, True])
def test_qk_norm_rope_q_scale(extras):
    """q_scale = 1 through the _qs entry is the plain entry bit for bit; q_scale = c: Q (and Qt) = ONE bf16 rounding of the
    fp32 q * c (relative L2 to the fp32 restatement <= 3e-3 = bf16 rounding, and strictly better than rounding twice), K, V^T,
    V, K^T unchanged bit for bit."""
    from mixgrpo_amd import ops
    from mixgrpo_amd._lib import lib, ptr, stream
    B, H, S, rows, s0 = 2, 3, 200, 150, 50
    Sp = 256
    qkv, wq, wk, cos, sin = _qkv_case(B, H, S, rows, s0, 5)
    Q1, K1, Vt1, ex1 = _run_fwd(ops, qkv, wq, wk, cos, sin, B, H, S, Sp, rows, s0, 1.0, extras)
    Q0, K0 = torch.zeros_like(Q1), torch.zeros_like(K1)
    Vt0 = torch.zeros_like(Vt1)
    ex0 = {k: torch.zeros_like(v) for k, v in ex1.items()}
    rc = lib().mgx_qk_norm_rope_fwd(ptr(qkv), qkv.shape[-1], ptr(wq), ptr(wk), ptr(cos), ptr(sin), ptr(Q0), ptr(K0), ptr(Vt0),
                                    ptr(ex0.get("V")), ptr(ex0.get("Qt")), ptr(ex0.get("Kt")), B, H, S, Sp, rows, s0, stream())
    assert rc == 0
    assert torch.equal(Q0, Q1) and torch.equal(K0, K1) and torch.equal(Vt0, Vt1)
    for k in ex1:
        assert torch.equal(ex0[k], ex1[k])
    Q2, K2, Vt2, ex2 = _run_fwd(ops, qkv, wq, wk, cos, sin, B, H, S, Sp, rows, s0, C, extras)
    assert torch.equal(K2, K1) and torch.equal(Vt2, Vt1)
    if extras:
        assert torch.equal(ex2["V"], ex1["V"]) and torch.equal(ex2["Kt"], ex1["Kt"])
        assert torch.equal(ex2["Qt"][..., s0:s0 + rows], Q2[:, :, s0:s0 + rows].transpose(-1, -2))
    # fp32 restatement of RMSNorm (eps 1e-6) + interleaved-pair RoPE on the q columns
    x = qkv[:, :H * 128].float().view(B, rows, H, 128).permute(0, 2, 1, 3)
    y = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * wq
    c_, s_ = cos[s0:s0 + rows], sin[s0:s0 + rows]
    ye, yo = y[..., 0::2], y[..., 1::2]
    ref = torch.stack([ye * c_[:, 0::2] - yo * s_[:, 0::2], yo * c_[:, 1::2] + ye * s_[:, 1::2]], -1).flatten(-2)
    e_once = rel_err(Q2[:, :, s0:s0 + rows], ref * C)
    e_twice = rel_err((Q1[:, :, s0:s0 + rows].float() * C).bfloat16(), ref * C)
    assert e_once < 3e-3 and e_once < e_twice
    assert not Q2[:, :, :s0].any() and not Q2[:, :, s0 + rows:].any()


def test_qk_norm_rope_bwd_q_scale_is_linear_in_dq():
    """dQ enters multiplied by q_scale and nothing else changes: the q columns of dqkv and the norm_q weight gradient are
    q_scale x the plain entry's (bf16 outputs: <= 4e-3 relative; fp32 weight gradient: 1e-5), the k / v columns and the
    norm_k gradient are the plain entry's bit for bit."""
    from mixgrpo_amd import ops
    B, H, S, rows, s0 = 2, 3, 200, 150, 50
    Sp = 256
    qkv, wq, wk, cos, sin = _qkv_case(B, H, S, rows, s0, 6)
    g = torch.Generator(device="cuda").manual_seed(7)
    dQ, dK, dV = (torch.randn(B, H, S, 128, device="cuda", generator=g).bfloat16() for _ in range(3))
    outs = []
    for qs in (1.0, C):
        dqkv = torch.zeros_like(qkv)
        gwq, gwk = torch.zeros(128, device="cuda"), torch.zeros(128, device="cuda")
        ops.qk_norm_rope_bwd(qkv, wq, wk, cos, sin, dQ, dK, dV, dqkv, gwq, gwk, B, H, S, Sp, rows, s0, q_scale=qs)
        outs.append((dqkv, gwq, gwk))
    (d1, gq1, gk1), (d2, gq2, gk2) = outs
    d = H * 128
    assert torch.equal(d1[:, d:], d2[:, d:]) and torch.equal(gk1, gk2)
    assert rel_err(d2[:, :d], d1[:, :d].float() * C) < 4e-3
    assert torch.allclose(gq2, gq1 * C, rtol=1e-5, atol=1e-6)


def _reference(q2, k, v):
    s = (q2.float() @ k.float().transpose(-1, -2)) * LN2
    B, H, S, _ = q2.shape
    return s, (torch.softmax(s, -1) @ v.float()).transpose(1, 2).reshape(B, S, H * 128)


@pytest.mark.parametrize("B,H,S,ldo_mult", [(1, 1, 256, 1), (2, 3, 768, 5), (1, 24, 1536, 1), (2, 3, 1100, 1)])
def test_attention_log2_vs_torch_and_vs_the_8_wave_kernel(B, H, S, ldo_mult, monkeypatch):
    """mgx_attn_fwd_log2 on Q2 = bf16(q c) against fp32 softmax(ln 2 * Q2 K^T) V of the same operands (<= 6e-3 relative L2,
    lse 1e-4), on the 64-query kernel (S % 256 == 0) and on the 8-wave kernel (MGX_ATTN_W64=0 / S = 1100 with padding); the two
    agree within bf16 rounding of O; columns beyond the head block stay untouched; and against mgx_attn_fwd(Q2, scale = ln 2),
    the same mathematics through the multiply-add softmax."""
    from mixgrpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(S + H)
    q, k, v = (torch.randn(B, H, S, 128, device="cuda", generator=g).bfloat16() for _ in range(3))
    q2 = (q.float() * C).bfloat16()
    Sp = (S + 63) // 64 * 64
    vt = torch.zeros(B, H, 128, Sp, device="cuda", dtype=torch.bfloat16)
    vt[..., :S] = v.transpose(-1, -2)
    ldo = H * 128 * ldo_mult
    outs = []
    for w64 in ("1", "0"):
        monkeypatch.setenv("MGX_ATTN_W64", w64)
        O = torch.zeros(B, S, ldo, device="cuda", dtype=torch.bfloat16)
        lse = torch.empty(B, H, S, device="cuda")
        ops.attn_fwd_log2(q2, k, vt, O, lse, B, H, S, Sp, ldo, S * ldo)
        outs.append((O, lse))
    monkeypatch.setenv("MGX_ATTN_W64", "1")
    O = torch.zeros(B, S, ldo, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, H, S, device="cuda")
    ops.attn_fwd(q2, k, vt, O, lse, B, H, S, Sp, ldo, S * ldo, LN2)
    outs.append((O, lse))
    s, ref = _reference(q2, k, v)
    for O, lse in outs:
        assert rel_err(O[:, :, :H * 128], ref) < 6e-3
        assert torch.allclose(lse, torch.logsumexp(s, -1), rtol=1e-4, atol=1e-4)
        assert not O[:, :, H * 128:].any()
    for O, lse in outs[1:]:
        assert rel_err(outs[0][0], O.float()) < 4e-3
        assert torch.allclose(outs[0][1], lse, rtol=1e-5, atol=2e-5)


def test_attention_log2_rescale_path():
    """tests/test_hip_mmdit.py::test_attention_fwd64_rescale_path on the accumulator-initialised kernel: its fix-up moves m by
    the tile's own maximum and rewrites the -m blocks, in a middle tile and in the last one, both chains."""
    from mixgrpo_amd import ops
    B, H, S = 1, 2, 1024
    g = torch.Generator(device="cuda").manual_seed(11)
    q, k, v = (torch.randn(B, H, S, 128, device="cuda", generator=g).bfloat16() for _ in range(3))
    k[:, :, 200] = (8 * q[:, :, 70].float()).bfloat16()
    k[:, :, S - 3] = (6 * q[:, :, 100].float()).bfloat16()
    k[:, :, 333] = (7 * q[:, :, 700].float()).bfloat16()
    q2 = (q.float() * C).bfloat16()
    O = torch.empty(B, S, H * 128, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, H, S, device="cuda")
    ops.attn_fwd_log2(q2, k, v.transpose(-1, -2).contiguous(), O, lse, B, H, S, S, H * 128, S * H * 128)
    s, ref = _reference(q2, k, v)
    assert torch.isfinite(O.float()).all()
    assert rel_err(O, ref) < 6e-3
    got = O.view(B, S, H, 128).permute(0, 2, 1, 3)
    refh = ref.view(B, S, H, 128).permute(0, 2, 1, 3)
    for row in (70, 100, 700):
        assert rel_err(got[:, :, row], refh[:, :, row]) < 6e-3
    assert torch.allclose(lse, torch.logsumexp(s, -1), rtol=1e-4, atol=1e-3)


def test_attention_log2_full_size_properties_and_backward():
    """BASELINE-size attention (H = 24, S = 4608) on the prescaled path: rows of P sum to one (V = 1 gives O = 1 exactly),
    sampled rows against fp32 torch, invariance under a joint permutation of keys and values; and the backward at
    scale = ln 2 on (Q2, Q2^T): sampled rows of dQ2 / dK / dV against fp32 autograd of softmax(ln 2 Q2 K^T) V on a 768-token
    problem (<= 1.5e-2), plus at full size the identity sum <q2, dq2> = sum <k, dk>."""
    from mixgrpo_amd import ops
    B, H, S = 1, 24, 4608
    g = torch.Generator(device="cuda").manual_seed(3)
    q, k, v = (torch.randn(B, H, S, 128, device="cuda", generator=g).bfloat16() for _ in range(3))
    q2 = (q.float() * C).bfloat16()

    def fwd(q_, k_, v_):
        O = torch.empty(B, S, H * 128, device="cuda", dtype=torch.bfloat16)
        lse = torch.empty(B, H, S, device="cuda")
        ops.attn_fwd_log2(q_, k_, v_.transpose(-1, -2).contiguous(), O, lse, B, H, S, S, H * 128, S * H * 128)
        return O.view(B, S, H, 128).permute(0, 2, 1, 3), lse

    O1, _ = fwd(q2, k, torch.ones_like(v))
    assert torch.equal(O1.float(), torch.ones_like(O1).float())
    O, lse = fwd(q2, k, v)
    rows = torch.randint(0, S, (96,), generator=torch.Generator().manual_seed(1)).cuda()
    s = (q2[:, :, rows].float() @ k.float().transpose(-1, -2)) * LN2
    assert rel_err(O[:, :, rows], torch.softmax(s, -1) @ v.float()) < 6e-3
    assert torch.allclose(lse[:, :, rows], torch.logsumexp(s, -1), rtol=1e-4, atol=1e-4)
    perm = torch.randperm(S, generator=torch.Generator().manual_seed(2)).cuda()
    Op, lsep = fwd(q2, k[:, :, perm].contiguous(), v[:, :, perm].contiguous())
    assert rel_err(Op, O.float()) < 4e-3 and torch.allclose(lsep, lse, rtol=1e-5, atol=1e-5)

    tr = lambda t: t.transpose(-1, -2).contiguous()
    do = torch.randn(B, S, H * 128, device="cuda", generator=g).bfloat16()
    Oc = O.permute(0, 2, 1, 3).reshape(B, S, H * 128).contiguous()
    dQ, dK, dV = (torch.empty_like(q) for _ in range(3))
    delta = torch.empty(B, H, S, device="cuda")
    dOt = torch.zeros(B, H, 128, S, device="cuda", dtype=torch.bfloat16)
    ops.attn_bwd(q2, k, v, tr(q2), tr(k), Oc, do, lse, delta, dOt, dQ, dK, dV, B, H, S, S, H * 128, S * H * 128, LN2)
    a = (q2.float() * dQ.float()).sum().item()
    b = (k.float() * dK.float()).sum().item()
    assert abs(a - b) <= 2e-2 * max(abs(a), abs(b), 1.0)

    B, H, S = 1, 3, 768
    q, k, v = (torch.randn(B, H, S, 128, device="cuda", generator=g).bfloat16() for _ in range(3))
    q2 = (q.float() * C).bfloat16()
    do = torch.randn(B, S, H * 128, device="cuda", generator=g).bfloat16()
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q2, k, v))
    o_ref = (torch.softmax((qf @ kf.transpose(-1, -2)) * LN2, -1) @ vf).transpose(1, 2).reshape(B, S, H * 128)
    o_ref.backward(do.float())
    O = torch.empty(B, S, H * 128, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, H, S, device="cuda")
    ops.attn_fwd_log2(q2, k, tr(v), O, lse, B, H, S, S, H * 128, S * H * 128)
    dQ, dK, dV = (torch.full_like(q, float("nan")) for _ in range(3))
    delta = torch.empty(B, H, S, device="cuda")
    dOt = torch.zeros(B, H, 128, S, device="cuda", dtype=torch.bfloat16)
    ops.attn_bwd(q2, k, v, tr(q2), tr(k), O, do, lse, delta, dOt, dQ, dK, dV, B, H, S, S, H * 128, S * H * 128, LN2)
    assert rel_err(dV, vf.grad) < 1e-2
    assert rel_err(dK, kf.grad) < 1.5e-2
    assert rel_err(dQ, qf.grad) < 1.5e-2


def test_model_with_and_without_prescaled_q_agree_with_the_oracle(monkeypatch):
    """The whole model, forward and parameter gradients, with MGX_ATTN_Q_PRESCALE on (default) and off: both within the
    test_hip_mmdit tolerances of the oracle, and within bf16 rounding of each other."""
    from oracle import mmdit as OM
    from mixgrpo_amd import ops
    import test_hip_mmdit as T
    B, hg, wg, L = 2, 8, 8, 64                       # S = 128: the 8-wave kernel; 64-query shapes are covered above
    res = []
    for on in (True, False):
        monkeypatch.setattr(ops, "Q_PRESCALE", on)
        ocfg, P, m = T.build_pair(T.small_cfg(2, 2))
        assert (m.q_scale() != 1.0) == on and (abs(m.attn_scale() - LN2) < 1e-12) == on
        x, ehs, pooled, ids, tids, t, gd = T.make_inputs(B, hg, wg, L, seed=3)
        R = torch.randn(B, hg * wg, 64, generator=torch.Generator().manual_seed(9))
        m.train()
        out = m(x.cuda(), ehs.cuda(), t.cuda(), gd.cuda(), tids.cuda(), pooled.cuda(), ids.cuda())[0]
        (out.float() * R.cuda()).sum().backward()
        res.append((out.detach().float().cpu(), torch.cat([m.store.view(m.store.g32, k).float().reshape(-1).cpu() for k in P])))
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref = OM.forward(Pg, ocfg, x, ehs.float(), t, gd.float(), tids, pooled.float(), ids)
    (ref * R).sum().backward()
    gref = torch.cat([Pg[k].grad.reshape(-1) for k in P])
    for out, g in res:
        assert T.rel_err(out, ref) < 1e-2
        assert torch.nn.functional.cosine_similarity(g, gref, dim=0).item() > 0.999
    assert T.rel_err(res[0][0], res[1][0]) < 8e-3
    assert torch.nn.functional.cosine_similarity(res[0][1], res[1][1], dim=0).item() > 0.9995


def test_forward_with_fused_projections_equals_the_norm_pass(monkeypatch):
    """The no-grad forward (rollout) three ways: (a) V^T written by mgx_linear_bf16_t and QK-norm / RoPE in the q | k projection's
    epilogue (mgx_linear_qk_norm_rope), (b) V^T direct + q | k-only norm pass (MGX_LINEAR_QKNORM=0), (c) the fused q | k | v
    projection + mgx_qk_norm_rope_fwd's transposing pass (MGX_LINEAR_VT=0), at a size where the persistent kernel takes the
    projections (FLUX width, one double + one single block, 8 x (256 + 1024) tokens), stream-K off: every output element is the
    same K-loop and the same norm arithmetic in the same order, so the outputs are equal BIT FOR BIT."""
    from mixgrpo_amd import ops
    from mixgrpo_amd.flux import FluxConfig, FluxTransformer2DModel
    cfg = dict(num_layers=1, num_single_layers=1, attention_head_dim=128, num_attention_heads=24, joint_attention_dim=64,
               pooled_projection_dim=32)
    torch.manual_seed(0)
    m = FluxTransformer2DModel(FluxConfig(**cfg), device="cuda")
    sd = {k: (torch.randn(v.shape, device="cuda") * (0.02 if v.dim() > 1 else 0.05) + (1.0 if "norm_" in k and v.dim() == 1 else 0.0))
          for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    m.eval()
    B, hg, wg, L = 8, 32, 32, 256
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, hg * wg, 64, generator=g).cuda()
    ehs = torch.randn(B, L, 64, generator=g).bfloat16().cuda()
    pooled = torch.randn(B, 32, generator=g).bfloat16().cuda()
    ids = torch.zeros(hg, wg, 3)
    ids[..., 1] += torch.arange(hg)[:, None]
    ids[..., 2] += torch.arange(wg)[None]
    ids = ids.reshape(-1, 3).cuda()
    t = torch.full((B,), 0.7).cuda()
    gd = torch.tensor([3.5]).bfloat16().cuda()
    monkeypatch.setattr(ops, "GEMM_STREAM_K", False)
    calls = {"vt": [], "qk": []}
    real_t, real_qk = ops.linear_t, ops.linear_qk_norm_rope
    monkeypatch.setattr(ops, "linear_t", lambda *a, **k: (calls["vt"].append(real_t(*a, **k)) or calls["vt"][-1]))
    monkeypatch.setattr(ops, "linear_qk_norm_rope", lambda *a, **k: (calls["qk"].append(real_qk(*a, **k)) or calls["qk"][-1]))
    outs = []
    for vt, qk in ((True, True), (True, False), (False, False)):
        monkeypatch.setattr(ops, "LINEAR_VT", vt)
        monkeypatch.setattr(ops, "LINEAR_QKNORM", qk)
        with torch.no_grad():
            outs.append(m._forward_nograd(x, ehs, t, gd, torch.zeros(L, 3).cuda(), pooled, ids).clone())
    # text stream: 8 x 256 tokens = 12 x 8 = 96 tiles of V^T, declined (plain path inside the same forward); image stream and the
    # single block's joint sequence: taken, both times; the fused q | k epilogue: taken where it was tried
    assert calls["vt"] == [False, True, True] * 2 and calls["qk"] == [True, True]
    assert torch.isfinite(outs[0].float()).all() and outs[0].float().abs().max() > 0
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[2])
