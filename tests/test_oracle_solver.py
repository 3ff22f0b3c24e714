"""Pins oracle/solver.py bit-exactly against vectors produced by the reference's own functions."""
from argparse import Namespace

import pytest
import torch

from helpers import assert_same, host_matches_fixture_host, load_golden
from oracle import solver as O

T_, M_ = load_golden("solver_steps")


EXACT = host_matches_fixture_host()


def eq(a, b):
    assert_same(a, b, exact=EXACT)


def test_sigma_schedules():
    for name, info in M_["schedules"].items():
        shift, T = name.split("_")
        shift, T = float(shift[5:]), int(T[1:])
        s = O.sd3_time_shift(shift, torch.linspace(1, 0, T + 1))
        eq(s, T_[f"sigma/{name}"])
        assert [int(x * 1000) for x in s] == info["timesteps"]


@pytest.mark.parametrize("case", [c for c in M_["cases"] if c["kind"] == "flow"], ids=lambda c: c["key"])
def test_flow_step(case):
    x, v = T_["in/x"], T_["in/v"]
    sig = T_[f"sigma/shift3.0_T{case['T']}"]  # stored schedule: torch.linspace is not bit-stable across CPUs
    k = case["key"]
    out = O.flow_grpo_step(v, x, case["eta"], sig, case["index"], None, determistic=case["det"], noise=T_[k + "/noise"])
    for nm, t in zip(("prev", "x0", "logp", "mean", "std"), out):
        eq(t, T_[f"{k}/{nm}"])
    if not case["det"]:
        rp = O.flow_grpo_step(v, x, case["eta"], sig, case["index"], out[0].clone())
        eq(rp[2], T_[f"{k}/replay_logp"])
        eq(rp[2], out[2])  # replay identity


def test_flow_step_rejects_generator_and_prev():
    x, v = T_["in/x"], T_["in/v"]
    sig = T_["sigma/shift3.0_T8"]
    with pytest.raises(ValueError):
        O.flow_grpo_step(v, x, 0.7, sig, 1, x, generator=torch.Generator())


@pytest.mark.parametrize("case", [c for c in M_["cases"] if c["kind"] == "dance"], ids=lambda c: c["key"])
def test_dance_step(case):
    x, v = T_["in/x"], T_["in/v"]
    sig = T_["sigma/shift3.0_T8"]
    k = case["key"]
    out = O.dance_grpo_step(v, x, case["eta"], sig, case["index"], None, True, case["sde"], noise=T_[k + "/noise"])
    for nm, t in zip(("prev", "x0", "logp"), out):
        eq(t, T_[f"{k}/{nm}"])
    rp = O.dance_grpo_step(v, x, case["eta"], sig, case["index"], out[0].clone(), True, True)
    eq(rp[2], T_[f"{k}/replay_logp_sde"])


@pytest.mark.parametrize("case", [c for c in M_["cases"] if c["kind"] == "dpm"], ids=lambda c: c["key"])
def test_dpm_chain(case):
    x = T_["in/x"]
    T = case["T"]
    sig = T_[f"sigma/shift3.0_T{T}"]
    a = Namespace(dpm_algorithm_type=case["algo"], dpm_solver_order=case["order"], dpm_solver_type=case["stype"])
    st = O.DPMState(order=case["order"])
    xs = x.clone()
    k = case["key"]
    for i in range(T):
        noise = T_[f"{k}/s{i}/noise"] if case["sde"] else None
        prev, x0, lp = O.dpm_step(a, T_[f"{k}/s{i}/v"], xs, i, sig[:-1], sig, dpm_state=st, variance_noise=noise,
                                  sde_solver=case["sde"])
        eq(prev, T_[f"{k}/s{i}/prev"])
        eq(lp, T_[f"{k}/s{i}/logp"])
        if i in (0, 3):
            eq(x0, T_[f"{k}/s{i}/x0"])
        xs = prev


def test_dpm_without_state():
    a = Namespace(dpm_algorithm_type="dpmsolver++", dpm_solver_order=2, dpm_solver_type="midpoint")
    sig = T_["sigma/shift3.0_T8"]
    prev, x0, lp = O.dpm_step(a, T_["in/v"], T_["in/x"], 3, sig[:-1], sig, dpm_state=None,
                              variance_noise=T_["dpm/nostate/noise"], sde_solver=True)
    eq(prev, T_["dpm/nostate/prev"])
    eq(lp, T_["dpm/nostate/logp"])


@pytest.mark.parametrize("algo", ["dpmsolver++", "dpmsolver"])
def test_dpm_without_state_gradient(algo):
    """The replay under dpm_apply_strategy="all" (reference train_grpo_flux.py:170-180): the state-less first-order SDE
    dpm_step's log-prob differentiates through prev_sample_mean (sampling_utils.py:376-383).  Fixture: the reference's own
    autograd gradient w.r.t. the bf16 model output."""
    a = Namespace(dpm_algorithm_type=algo, dpm_solver_order=2, dpm_solver_type="midpoint")
    sig = T_["sigma/shift3.0_T8"]
    v = T_["in/v"].clone().requires_grad_(True)
    _, _, lp = O.dpm_step(a, v, T_["in/x"], 3, sig[:-1], sig, dpm_state=None, variance_noise=T_["dpm/nostate/noise"],
                          sde_solver=True)
    eq(lp.detach(), T_[f"dpm/nostate_grad/{algo}/logp"])
    (lp * T_[f"dpm/nostate_grad/{algo}/upstream"]).sum().backward()
    assert v.grad.abs().max() > 0
    eq(v.grad, T_[f"dpm/nostate_grad/{algo}/grad_v"])
