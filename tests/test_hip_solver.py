"""GPU parity: HIP solver kernels (through the C ABI) vs the CPU oracle and the reference's golden vectors.

Bar: elementwise fp32 outputs (prev_sample, x0, mean, dv) bit-exact; log-probs (a mean over N*64 elements
whose summation order differs on the GPU) within 2e-6 relative -- the north star allows 1e-3.
"""
from argparse import Namespace

import pytest
import torch

from helpers import assert_same, load_golden
from oracle import rollout as OR
from oracle import solver as O
from toy_model import ElementwiseToy

pytestmark = pytest.mark.gpu
T_, M_ = load_golden("solver_steps")
LP_RTOL = 2e-6


def dev(t):
    return t.cuda() if t is not None else None


def eq(a, b):
    a = a.detach().cpu()
    b = b.detach()
    assert a.dtype == b.dtype and a.shape == b.shape
    assert torch.equal(torch.nan_to_num(a, nan=777.0), torch.nan_to_num(b, nan=777.0)), (a - b).abs().max()


def close_lp(a, b, rtol=LP_RTOL):
    a = a.detach().cpu()
    assert a.shape == b.shape
    fin = torch.isfinite(b)
    assert torch.equal(torch.isnan(a), torch.isnan(b))
    assert torch.allclose(a[fin], b[fin], rtol=rtol, atol=1e-7), (a, b)


@pytest.mark.parametrize("case", [c for c in M_["cases"] if c["kind"] == "flow"], ids=lambda c: c["key"])
def test_flow_step_golden(case):
    from mixgrpo_amd import sampling_utils as SU
    x, v = T_["in/x"], T_["in/v"]
    sig = T_[f"sigma/shift3.0_T{case['T']}"]  # stored schedule: torch.linspace is not bit-stable across CPUs
    k = case["key"]
    out = SU.flow_grpo_step(dev(v), dev(x), case["eta"], sig, case["index"], None, determistic=case["det"],
                            noise=dev(T_[k + "/noise"]))
    ora = O.flow_grpo_step(v, x, case["eta"], sig, case["index"], None, determistic=case["det"], noise=T_[k + "/noise"])
    for j, nm in enumerate(("prev", "x0", "logp", "mean", "std")):
        if nm == "logp":
            close_lp(out[2], ora[2])
        else:
            eq(out[j], ora[j])                                   # same host scalars -> bit-exact
        assert_same(out[j], T_[f"{k}/{nm}"], exact=False)        # reference vector from another host: few ulps
    if not case["det"]:
        rp = SU.flow_grpo_step(dev(v), dev(x), case["eta"], sig, case["index"], out[0].clone())
        assert_same(rp[2], T_[f"{k}/replay_logp"], exact=False)
        assert torch.equal(rp[2], out[2])   # replay identity on the device path too


def test_flow_step_errors():
    from mixgrpo_amd import sampling_utils as SU
    sig = T_["sigma/shift3.0_T8"]
    x, v = dev(T_["in/x"]), dev(T_["in/v"])
    with pytest.raises(ValueError):
        SU.flow_grpo_step(v, x, 0.7, sig, 1, x, generator=torch.Generator(device="cuda"))


@pytest.mark.parametrize("T,index", [(8, 1), (8, 3), (25, 2), (25, 12)])
def test_flow_replay_backward_vs_oracle_autograd(T, index):
    from mixgrpo_amd import sampling_utils as SU
    g = torch.Generator().manual_seed(5 + index)
    x = torch.randn(3, 64, 64, generator=g)
    v = torch.randn(3, 64, 64, generator=g).bfloat16()
    sig = SU.sd3_time_shift(3.0, torch.linspace(1, 0, T + 1))
    prev = O.flow_grpo_step(v, x, 0.7, sig, index, None, noise=torch.randn(3, 64, 64, generator=g).bfloat16())[0]
    w = torch.tensor([0.3, -1.7, 2.5])
    v_o = v.clone().requires_grad_(True)
    lp_o = O.flow_grpo_step(v_o, x, 0.7, sig, index, prev)[2]
    (lp_o * w).sum().backward()
    v_h = dev(v).requires_grad_(True)
    lp_h = SU.flow_grpo_step(v_h, dev(x), 0.7, sig, index, dev(prev), want_x0=False, want_mean=False)[2]
    (lp_h * dev(w)).sum().backward()
    close_lp(lp_h, lp_o.detach())
    eq(v_h.grad, v_o.grad)


@pytest.mark.parametrize("case", [c for c in M_["cases"] if c["kind"] == "dance"], ids=lambda c: c["key"])
def test_dance_step_golden(case):
    from mixgrpo_amd import sampling_utils as SU
    x, v = T_["in/x"], T_["in/v"]
    sig = T_["sigma/shift3.0_T8"]
    k = case["key"]
    out = SU.dance_grpo_step(dev(v), dev(x), case["eta"], sig, case["index"], None, True, case["sde"],
                             noise=dev(T_[k + "/noise"]))
    ora = O.dance_grpo_step(v, x, case["eta"], sig, case["index"], None, True, case["sde"], noise=T_[k + "/noise"])
    eq(out[0], ora[0])
    eq(out[1], ora[1])
    close_lp(out[2], ora[2])
    for j, nm in enumerate(("prev", "x0", "logp")):
        assert_same(out[j], T_[f"{k}/{nm}"], exact=False)
    rp = SU.dance_grpo_step(dev(v), dev(x), case["eta"], sig, case["index"], out[0].clone(), True, True)
    assert_same(rp[2], T_[f"{k}/replay_logp_sde"], exact=False)
    mean, x0 = SU.dance_grpo_step(dev(v), dev(x), case["eta"], sig, case["index"], None, False, case["sde"])
    om, ox0 = O.dance_grpo_step(v, x, case["eta"], sig, case["index"], None, False, case["sde"])
    eq(mean, om)
    eq(x0, ox0)


@pytest.mark.parametrize("sde", [False, True])
def test_dance_replay_backward_vs_oracle_autograd(sde):
    from mixgrpo_amd import sampling_utils as SU
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 32, 64, generator=g)
    v = torch.randn(2, 32, 64, generator=g).bfloat16()
    sig = T_["sigma/shift3.0_T8"]
    prev = O.dance_grpo_step(v, x, 0.3, sig, 2, None, True, True, noise=torch.randn(2, 32, 64, generator=g))[0]
    w = torch.tensor([1.3, -0.7])
    v_o = v.clone().requires_grad_(True)
    lp_o = O.dance_grpo_step(v_o, x, 0.3, sig, 2, prev, True, sde)[2]
    (lp_o * w).sum().backward()
    v_h = dev(v).requires_grad_(True)
    lp_h = SU.dance_grpo_step(v_h, dev(x), 0.3, sig, 2, dev(prev), True, sde)[2]
    (lp_h * dev(w)).sum().backward()
    close_lp(lp_h, lp_o.detach())
    eq(v_h.grad, v_o.grad)


@pytest.mark.parametrize("case", [c for c in M_["cases"] if c["kind"] == "dpm"], ids=lambda c: c["key"])
def test_dpm_chain_golden(case):
    from mixgrpo_amd import sampling_utils as SU
    T = case["T"]
    sig = T_[f"sigma/shift3.0_T{T}"]
    a = Namespace(dpm_algorithm_type=case["algo"], dpm_solver_order=case["order"], dpm_solver_type=case["stype"])
    st, ost = SU.DPMState(order=case["order"]), O.DPMState(order=case["order"])
    xs, oxs = dev(T_["in/x"]), T_["in/x"]
    k = case["key"]
    for i in range(T):
        noise = T_[f"{k}/s{i}/noise"] if case["sde"] else None
        prev, x0, lp = SU.dpm_step(a, dev(T_[f"{k}/s{i}/v"]), xs, i, sig[:-1], sig, dpm_state=st,
                                   variance_noise=dev(noise), sde_solver=case["sde"])
        oprev, ox0, olp = O.dpm_step(a, T_[f"{k}/s{i}/v"], oxs, i, sig[:-1], sig, dpm_state=ost, variance_noise=noise,
                                     sde_solver=case["sde"])
        eq(prev, oprev)
        eq(x0, ox0)
        close_lp(lp, olp, rtol=1e-5)
        assert_same(prev, T_[f"{k}/s{i}/prev"], exact=False, rtol=1e-5, atol=1e-5)
        assert_same(lp, T_[f"{k}/s{i}/logp"], exact=False, rtol=1e-4, atol=1e-5)
        xs, oxs = prev, oprev


def test_dpm_without_state_and_unreachable_order():
    from mixgrpo_amd import sampling_utils as SU
    a = Namespace(dpm_algorithm_type="dpmsolver++", dpm_solver_order=2, dpm_solver_type="midpoint")
    sig = T_["sigma/shift3.0_T8"]
    prev, x0, lp = SU.dpm_step(a, dev(T_["in/v"]), dev(T_["in/x"]), 3, sig[:-1], sig, dpm_state=None,
                               variance_noise=dev(T_["dpm/nostate/noise"]), sde_solver=True)
    assert_same(prev, T_["dpm/nostate/prev"], exact=False, rtol=1e-5, atol=1e-5)
    assert_same(lp, T_["dpm/nostate/logp"], exact=False, rtol=1e-4, atol=1e-5)
    with pytest.raises(NotImplementedError):
        SU.dpm_coeffs("dpmsolver", "midpoint", 3, sig, 3, False)


R_, RM_ = load_golden("rollout")


@pytest.mark.parametrize("case", RM_["cases"], ids=lambda c: c["tag"])
def test_rollout_vs_oracle(case):
    """Whole rollouts with an elementwise toy model (bit-identical on CPU and GPU) and injected noise."""
    from mixgrpo_amd import sampling_utils as SU
    a = Namespace(**case["args"])
    T = case["T"]
    sig = SU.sd3_time_shift(a.shift, torch.linspace(1, 0, T + 1))
    det = [i not in case["window"] for i in range(T)]
    z0 = torch.cat([R_["in/z0"], (R_["in/z0"].float() * 0.5 + 0.25).bfloat16()], 0)   # batch of 2
    ehs, pooled = R_["in/ehs"].repeat(2, 1, 1), R_["in/pooled"].repeat(2, 1)
    text_ids, ids = R_["in/text_ids"].repeat(2, 1), R_["in/img_ids"]
    g = torch.Generator().manual_seed(99)
    if "dpmsolver" in a.dpm_algorithm_type and a.dpm_apply_strategy == "all":
        ndt = torch.float32       # dpm_step draws fp32 noise (reference :319-321)
    elif a.flow_grpo_sampling:
        ndt = torch.bfloat16      # flow_grpo_step draws in model_output.dtype (:189-194)
    else:
        ndt = torch.float32       # dance_grpo_step: randn_like(fp32 mean) (:238)
    noises = [torch.randn(z0.shape, generator=g).to(ndt) for _ in range(T)]
    m_cpu, m_gpu = ElementwiseToy(), ElementwiseToy().cuda()
    with torch.no_grad():
        oz, olat, oall, olp = OR.run_sample_step(a, z0, range(T), sig, m_cpu, ehs, pooled, text_ids[:1], ids, True, det,
                                                 noises=[n.clone() for n in noises])
        hz, hlat, hall, hlp = SU.run_sample_step(a, dev(z0), range(T), sig, m_gpu, dev(ehs), dev(pooled),
                                                 dev(text_ids[:1]), dev(ids), True, det,
                                                 noises=[dev(n) for n in noises])
    eq(hall.contiguous(), oall)
    eq(hz, oz)
    eq(hlat, olat)
    close_lp(hlp.contiguous(), olp, rtol=1e-5)


def test_full_size_properties():
    """BASELINE size (B=8, N=4096, C=64): replay identity, ODE step has zero-noise dependence, pack round trip."""
    from mixgrpo_amd import sampling_utils as SU
    from mixgrpo_amd import latents as L
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(8, 4096, 64, device="cuda", generator=g)
    v = torch.randn(8, 4096, 64, device="cuda", generator=g).bfloat16()
    sig = SU.sd3_time_shift(3.0, torch.linspace(1, 0, 26))
    n1 = torch.randn(8, 4096, 64, device="cuda", generator=g).bfloat16()
    prev, _, lp, mean, sd = SU.flow_grpo_step(v, x, 0.7, sig, 2, None, noise=n1)
    lp2 = SU.flow_grpo_step(v, x, 0.7, sig, 2, prev, want_x0=False, want_mean=False)[2]
    assert torch.equal(lp, lp2)
    # log-prob of the draw equals the closed form in terms of the (bf16-rounded) injected noise
    d = (prev - mean)
    ref = (-(d.double() ** 2) / (2 * sd.double() ** 2) - torch.log(sd.double()) - 0.9189385332046727).mean(dim=(1, 2))
    assert torch.allclose(lp.double(), ref, rtol=1e-6)
    p_det1 = SU.flow_grpo_step(v, x, 0.7, sig, 2, None, determistic=True, noise=n1)[0]
    p_det2 = SU.flow_grpo_step(v, x, 0.7, sig, 2, None, determistic=True, noise=torch.zeros_like(n1))[0]
    assert torch.equal(p_det1, p_det2)
    lat = torch.randn(8, 16, 128, 128, device="cuda", generator=g).bfloat16()
    packed = L.pack_latents(lat, 8, 16, 128, 128)
    assert packed.shape == (8, 4096, 64)
    assert torch.equal(L.unpack_latents(packed, 1024, 1024, 8), lat)
    ref_pack = lat.view(8, 16, 64, 2, 64, 2).permute(0, 2, 4, 1, 3, 5).reshape(8, 4096, 64)
    assert torch.equal(packed, ref_pack)


@pytest.mark.parametrize("window,kw", [([3, 4], {}), ([0, 1], {}), ([2, 3], dict(flow_grpo_sampling=False, eta=0.3)),
                                       ([1, 2], dict(dpm_algorithm_type="dpmsolver++", dpm_apply_strategy="post",
                                                     dpm_post_compress_ratio=0.5, dpm_solver_order=2,
                                                     dpm_solver_type="midpoint"))])
def test_shared_prefix_rollout_is_bit_identical(window, kw):
    """Group rows that start identical: running the pre-window steps once (batch 1) and broadcasting must give exactly
    the rollout of the full batch."""
    from mixgrpo_amd import sampling_utils as SU
    base = dict(dpm_algorithm_type="null", dpm_apply_strategy="post", dpm_post_compress_ratio=0.4, dpm_solver_order=2,
                dpm_solver_type="midpoint", sample_strategy="progressive", shift=3.0, flow_grpo_sampling=True, eta=0.7,
                drop_last_sample=False)
    base.update(kw)
    a = Namespace(**base)
    T, B = 8, 4
    sig = SU.sd3_time_shift(a.shift, torch.linspace(1, 0, T + 1))
    det = [i not in window for i in range(T)]
    z0 = R_["in/z0"].repeat(B, 1, 1)
    ehs, pooled = R_["in/ehs"].repeat(B, 1, 1), R_["in/pooled"].repeat(B, 1)
    g = torch.Generator().manual_seed(3)
    ndt = torch.bfloat16 if a.flow_grpo_sampling else torch.float32
    noises = [torch.randn(z0.shape, generator=g).to(ndt) for _ in range(T)]
    m = ElementwiseToy().cuda()
    outs = []
    for shared in (False, True):
        with torch.no_grad():
            outs.append(SU.run_sample_step(a, dev(z0), range(T), sig, m, dev(ehs), dev(pooled), dev(R_["in/text_ids"]),
                                           dev(R_["in/img_ids"]), True, det, noises=[dev(n) for n in noises],
                                           shared_rows=shared))
    for x, y in zip(outs[0], outs[1]):
        assert torch.equal(torch.nan_to_num(x.contiguous(), nan=7.0), torch.nan_to_num(y.contiguous(), nan=7.0))


@pytest.mark.parametrize("algo", ["dpmsolver++", "dpmsolver"])
def test_dpm_replay_gradient(algo):
    """`mgx_dpm_step_bwd`: d log_prob / d model_output of the state-less first-order SDE dpm_step, the training replay under
    dpm_apply_strategy="all" (reference train_grpo_flux.py:170-180, sampling_utils.py:376-383) -- against the oracle's
    autograd on the same host (elementwise: bit-exact) and the reference's own gradient vector."""
    from mixgrpo_amd import sampling_utils as SU
    a = Namespace(dpm_algorithm_type=algo, dpm_solver_order=2, dpm_solver_type="midpoint")
    sig = T_["sigma/shift3.0_T8"]
    x, noise, up = T_["in/x"], T_["dpm/nostate/noise"], T_[f"dpm/nostate_grad/{algo}/upstream"]
    vo = T_["in/v"].clone().requires_grad_(True)
    po, _, lo = O.dpm_step(a, vo, x, 3, sig[:-1], sig, dpm_state=None, variance_noise=noise, sde_solver=True)
    (lo * up).sum().backward()
    vp = T_["in/v"].cuda().requires_grad_(True)
    pp, _, lp = SU.dpm_step(a, vp, dev(x), 3, sig[:-1], sig, dpm_state=None, variance_noise=dev(noise), sde_solver=True)
    assert lp.requires_grad
    (lp * up.cuda()).sum().backward()
    eq(pp, po.detach())
    close_lp(lp, lo.detach())
    assert vp.grad.dtype == torch.bfloat16 and vp.grad.abs().max() > 0
    eq(vp.grad, vo.grad)
    assert_same(vp.grad, T_[f"dpm/nostate_grad/{algo}/grad_v"], exact=False, rtol=1e-2, atol=1e-9)   # bf16 gradient: 1 ulp of bf16
    # without grad (the rollout) the same call still returns plain tensors
    with torch.no_grad():
        assert not SU.dpm_step(a, vp, dev(x), 3, sig[:-1], sig, dpm_state=None, variance_noise=dev(noise),
                               sde_solver=True)[2].requires_grad
