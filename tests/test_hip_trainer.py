"""GPU parity of the rollout-and-update engine (mixgrpo_amd.train_grpo_flux.train_one_step) against the CPU oracle
(oracle/trainer.py, itself pinned bit-exactly to the reference's train_one_step), with an elementwise toy velocity
model (bit-identical on CPU and GPU) and injected noise.  Elementwise results (latents) are bit-exact; log-probs,
advantages and losses (reductions) are compared at 1e-5 relative; the north star's bar is 1e-3."""
import copy
import random
from argparse import Namespace

import pytest
import torch

from oracle import trainer as OT
from toy_model import ElementwiseToy

pytestmark = pytest.mark.gpu


class _Sched:
    def step(self):
        pass


def base_args(**kw):
    a = dict(w=64, h=48, t=1, sampling_steps=8, shift=3.0, init_same_noise=True, training_strategy="part",
             output_dir="/tmp/x", experiment_name="t", reward_model="toy", multi_reward_mix="advantage_aggr",
             use_group=True, num_generations=4, trimmed_ratio=0.0, advantage_rerange_strategy="null", clip_range=1e-4,
             adv_clip_max=5.0, kl_coeff=0.0, gradient_accumulation_steps=2, frozen_init_timesteps=-1,
             timestep_fraction=1.0, dpm_algorithm_type="null", dpm_apply_strategy="post", dpm_post_compress_ratio=0.4,
             dpm_solver_order=2, dpm_solver_type="midpoint", sample_strategy="progressive", flow_grpo_sampling=True,
             eta=0.7, drop_last_sample=False, rollout_batch=0, train_microbatch=0)
    a.update(kw)
    return Namespace(**a)


CASES = [
    ("single_head", dict(), {"A": [0.1, 0.2, 0.3, 0.4]}, {"A": 1.0}, [2, 3]),
    ("multi_head_kl", dict(kl_coeff=0.01), {"A": [0.1, 0.5, 0.3, 0.9], "B": [2.0, 1.0, 4.0, 3.0]}, {"A": 1.0, "B": 0.5}, [0, 1]),
    ("trimmed", dict(num_generations=6, trimmed_ratio=0.25, gradient_accumulation_steps=3),
     {"A": [0.9, 0.2, 0.35, 0.4, 0.1, 0.77]}, {"A": 1.0}, [1, 2]),
    ("reward_aggr", dict(multi_reward_mix="reward_aggr"), {"A": [0.3, 0.1, 0.8, 0.4]}, {"A": 1.0}, [2, 3]),
    ("const_reward", dict(), {"A": [0.5] * 4}, {"A": 1.0}, [2, 3]),
    ("leftover", dict(num_generations=5, gradient_accumulation_steps=2), {"A": [0.3, 0.1, 0.8, 0.4, 0.6]}, {"A": 1.0}, [3, 4]),
    ("flash_post", dict(sampling_steps=12, dpm_algorithm_type="dpmsolver++", dpm_post_compress_ratio=0.4),
     {"A": [0.4, 0.2, 0.3, 0.1]}, {"A": 1.0}, [0, 1]),
    ("microbatched", dict(rollout_batch=2, train_microbatch=3), {"A": [0.1, 0.2, 0.3, 0.4]}, {"A": 1.0}, [2, 3]),
    # BASELINE.json configs[2] / configs[4] shapes of the schedule (bf16 attention): three reward heads with weights 1.0
    # on a group of 12, and a 50-step sampler with a 4-step window in the middle of the trajectory on a group of 16
    ("three_heads_g12", dict(num_generations=12, gradient_accumulation_steps=3),
     {"HPS": [0.1 * i for i in range(12)], "IR": [((7 * i) % 12) / 12 for i in range(12)],
      "Pick": [((5 * i + 3) % 12) / 6 for i in range(12)]}, {"HPS": 1.0, "IR": 1.0, "Pick": 1.0}, [0, 1, 2, 3]),
    ("sliding50_g16", dict(sampling_steps=50, num_generations=16, gradient_accumulation_steps=3),
     {"A": [((11 * i + 2) % 16) / 16 for i in range(16)]}, {"A": 1.0}, [10, 11, 12, 13]),
    # reference branches with pinned oracle fixtures (tests/golden/trainer.json: balance, dance_all) on the HIP path:
    # advantage_rerange_strategy (reward_model/utils.py:18-48 via train_grpo_flux.py:524-532; python `random`, seeded on
    # both sides; a zero advantage drops its sample) and training_strategy="all" (:503-522,552: every step SDE, DanceGRPO
    # solver, per-sample torch.randperm of the T-1 kept transitions, the first int((T-1)*timestep_fraction) of them trained)
    ("balance", dict(num_generations=6, advantage_rerange_strategy="balance"),
     {"A": [0.9, 0.2, 0.35, 0.4, 0.1, 0.77]}, {"A": 1.0}, [2, 3]),
    ("balance_drops_zero_adv", dict(num_generations=5, advantage_rerange_strategy="balance", gradient_accumulation_steps=2),
     {"A": [0.25, 0.5, 0.75, 0.5, 0.5]}, {"A": 1.0}, [2, 3]),
    ("rerange_random", dict(num_generations=6, advantage_rerange_strategy="random", gradient_accumulation_steps=3),
     {"A": [0.9, 0.2, 0.35, 0.4, 0.1, 0.77]}, {"A": 1.0}, [1, 2]),
    ("dance_all", dict(training_strategy="all", flow_grpo_sampling=False, timestep_fraction=0.6, kl_coeff=0.01),
     {"A": [0.1, 0.2, 0.3, 0.4]}, {"A": 1.0}, [2, 3]),
    ("flow_all_frozen", dict(training_strategy="all", frozen_init_timesteps=3, num_generations=5),
     {"A": [0.3, 0.1, 0.8, 0.4, 0.6]}, {"A": 1.0}, [0, 1]),
    # DPM-Solver++ on every step (dpm_apply_strategy="all"; pinned oracle fixture `dpm_all`): SDE dpm_step in the window
    # during the rollout, and a state-less first-order SDE dpm_step in the replay, whose log-prob carries a gradient through
    # its mean (train_grpo_flux.py:170-180).  The replay's noise comes from a freshly default-seeded generator in the
    # reference -- the same tensor for every pair; injected here (CPU and GPU generators differ)
    ("dpm_all", dict(dpm_algorithm_type="dpmsolver++", dpm_apply_strategy="all", kl_coeff=0.01),
     {"A": [0.4, 0.3, 0.1, 0.2]}, {"A": 1.0}, [2, 3]),
]


@pytest.mark.parametrize("tag,kw,rewards,weights,window", CASES, ids=[c[0] for c in CASES])
def test_train_one_step_vs_oracle(tag, kw, rewards, weights, window):
    from mixgrpo_amd import train_grpo_flux as TG
    a = base_args(**kw)
    G, T = a.num_generations, a.sampling_steps
    lh, lw = a.h // 8, a.w // 8
    N = (lh // 2) * (lw // 2)
    g = torch.Generator().manual_seed(5)
    inj = {"x_T": torch.randn(1, 16, lh, lw, generator=g).bfloat16(),
           "steps": [torch.randn(G, N, 64, generator=g).bfloat16() for _ in range(T)]}
    dpm_all = a.dpm_apply_strategy == "all" and "dpmsolver" in a.dpm_algorithm_type
    if dpm_all or not a.flow_grpo_sampling:
        inj["steps"] = [n.float() for n in inj["steps"]]    # dpm_step / dance_grpo_step draw fp32 noise (sampling_utils.py:319,237)
    if dpm_all:
        inj["dpm_replay"] = torch.randn((1, N, 64), generator=torch.Generator())   # what the oracle's fresh generator draws
    ehs = (0.1 * torch.randn(1, 8, 32, generator=g)).bfloat16()
    pooled = torch.randn(1, 16, generator=g).bfloat16()
    text_ids = torch.zeros(1, 3)
    heads = list(rewards)

    def o_reward(i, latents):
        rd = {h: [float(rewards[h][i])] for h in heads}
        return [sum(weights[h] * rd[h][0] for h in heads)], rd

    def p_reward(latents, captions):
        n = latents.shape[0]
        rd = {h: [float(rewards[h][i]) for i in range(n)] for h in heads}
        return [sum(weights[h] * rd[h][i] for h in heads) for i in range(n)], rd

    mo, mp = ElementwiseToy(), ElementwiseToy().cuda()
    oo = torch.optim.AdamW(mo.parameters(), lr=1e-2, betas=(0.9, 0.999), weight_decay=1e-4, eps=1e-8)
    po = torch.optim.AdamW(mp.parameters(), lr=1e-2, betas=(0.9, 0.999), weight_decay=1e-4, eps=1e-8)
    tro, trp = {}, {}
    torch.manual_seed(714)                       # host RNG consumers on the path: torch.randperm (strategy "all") ...
    random.seed(714)                             # ... and python `random` (balance_pos_neg); same stream on both sides
    ro = OT.train_one_step(a, mo, oo, _Sched(), (ehs, pooled, text_ids, ["p"]), o_reward, weights, window, 1.0, trace=tro,
                           injected=inj)
    ap = copy.copy(a)
    ap.injected_noise = inj
    torch.manual_seed(714)
    random.seed(714)
    rp = TG.train_one_step(ap, torch.device("cuda"), mp, None, p_reward, po, _Sched(),
                           iter([(ehs, pooled, text_ids, ["p"])]), None, 1.0, window, 0, weights, trace=trp)
    assert torch.allclose(trp["advantages"].cpu(), tro["advantages"], rtol=1e-5, atol=1e-6)
    lo, lp = tro["log_probs"], trp["log_probs"].cpu()
    fin = torch.isfinite(lo)
    assert torch.equal(torch.isfinite(lp), fin)
    assert torch.allclose(lp[fin], lo[fin], rtol=1e-5, atol=1e-6)
    for k in (0, 2, 3, 4):
        assert rp[k] == pytest.approx(ro[k], rel=2e-4, abs=3e-6), (k, rp, ro)  # sums of cancelling +-A*ratio terms of size 1/denom
    if ro[1] is None:
        assert rp[1] is None
    else:
        assert rp[1] == pytest.approx(ro[1], rel=2e-3, abs=1e-7)
    assert rp[5] == ro[5] or rp[5] == pytest.approx(ro[5], rel=1e-6)
    assert mp.a.item() == pytest.approx(mo.a.item(), rel=1e-4)


def test_skip_dead_backward_changes_nothing_but_work():
    """`args.skip_dead_backward`: the backward passes of the G % accum leftover samples (gradients the reference computes
    and then discards, train_grpo_flux.py:360,605-609) are not executed; every returned value and the updated weights
    must be identical to the default, reference-faithful run."""
    from mixgrpo_amd import train_grpo_flux as TG
    rewards = [0.3, 0.1, 0.8, 0.4, 0.6]
    a = base_args(num_generations=5, gradient_accumulation_steps=2)
    G, T = a.num_generations, a.sampling_steps
    lh, lw = a.h // 8, a.w // 8
    N = (lh // 2) * (lw // 2)
    g = torch.Generator().manual_seed(5)
    inj = {"x_T": torch.randn(1, 16, lh, lw, generator=g).bfloat16(),
           "steps": [torch.randn(G, N, 64, generator=g).bfloat16() for _ in range(T)]}
    batch = ((0.1 * torch.randn(1, 8, 32, generator=g)).bfloat16(), torch.randn(1, 16, generator=g).bfloat16(),
             torch.zeros(1, 3), ["p"])

    def reward(latents, captions):
        n = latents.shape[0]
        return [rewards[i] for i in range(n)], {"A": [rewards[i] for i in range(n)]}

    outs = []
    for skip in (False, True):
        m = ElementwiseToy().cuda()
        opt = torch.optim.AdamW(m.parameters(), lr=1e-2, betas=(0.9, 0.999), weight_decay=1e-4, eps=1e-8)
        ap = copy.copy(a)
        ap.injected_noise = inj
        ap.skip_dead_backward = skip
        calls = {"bwd": 0}
        h = m.a.register_hook(lambda gr: calls.__setitem__("bwd", calls["bwd"] + 1))
        res = TG.train_one_step(ap, torch.device("cuda"), m, None, reward, opt, _Sched(), iter([batch]), None, 1.0, [3, 4],
                                0, {"A": 1.0})
        h.remove()
        outs.append((res, m.a.item(), calls["bwd"]))
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1]
    assert outs[1][2] < outs[0][2]                      # fewer backward passes really ran


def _nogroup_case(B, rewards):
    """use_group=False (reference train_grpo_flux.py:494-499): a batch of B prompts, one sample each, advantages
    normalised by the mean / unbiased std of the GATHERED rewards -> `mgx_global_advantage`."""
    from mixgrpo_amd import train_grpo_flux as TG
    a = base_args(use_group=False, multi_reward_mix="reward_aggr", num_generations=1, gradient_accumulation_steps=2)
    T = a.sampling_steps
    lh, lw = a.h // 8, a.w // 8
    N = (lh // 2) * (lw // 2)
    g = torch.Generator().manual_seed(9)
    inj = {"x_T": torch.randn(1, 16, lh, lw, generator=g).bfloat16(),
           "steps": [torch.randn(B, N, 64, generator=g).bfloat16() for _ in range(T)]}
    ehs = (0.1 * torch.randn(B, 8, 32, generator=g)).bfloat16()
    pooled = torch.randn(B, 16, generator=g).bfloat16()
    text_ids = torch.zeros(B, 3)
    caps = [f"p{i}" for i in range(B)]
    mo, mp = ElementwiseToy(), ElementwiseToy().cuda()
    oo = torch.optim.AdamW(mo.parameters(), lr=1e-2, betas=(0.9, 0.999), weight_decay=1e-4, eps=1e-8)
    po = torch.optim.AdamW(mp.parameters(), lr=1e-2, betas=(0.9, 0.999), weight_decay=1e-4, eps=1e-8)
    tro, trp = {}, {}
    ro = OT.train_one_step(a, mo, oo, _Sched(), (ehs, pooled, text_ids, caps),
                           lambda i, lat: ([rewards[i]], {"A": [rewards[i]]}), {"A": 1.0}, [2, 3], 1.0, trace=tro, injected=inj)
    ap = copy.copy(a)
    ap.injected_noise = inj
    rp = TG.train_one_step(ap, torch.device("cuda"), mp, None,
                           lambda lat, cap: ([rewards[i] for i in range(lat.shape[0])], {"A": [rewards[i] for i in range(lat.shape[0])]}),
                           po, _Sched(), iter([(ehs, pooled, text_ids, caps)]), None, 1.0, [2, 3], 0, {"A": 1.0}, trace=trp)
    return ro, rp, tro, trp, mo, mp


def test_nogroup_global_advantage_vs_oracle():
    rewards = [0.3, 0.1, 0.8, 0.4]
    ro, rp, tro, trp, mo, mp = _nogroup_case(4, rewards)
    r = torch.tensor(rewards)
    assert torch.allclose(tro["advantages"], (r - r.mean()) / (r.std() + 1e-8))          # the reference formula (:498)
    assert torch.allclose(trp["advantages"].cpu(), tro["advantages"], rtol=1e-5, atol=1e-6)
    fin = torch.isfinite(tro["log_probs"])
    assert torch.allclose(trp["log_probs"].cpu()[fin], tro["log_probs"][fin], rtol=1e-5, atol=1e-6)
    for k in (0, 2, 3, 4):
        assert rp[k] == pytest.approx(ro[k], rel=2e-4, abs=3e-6), (k, rp, ro)
    assert rp[1] == pytest.approx(ro[1], rel=2e-3, abs=1e-7)
    assert rp[5] == pytest.approx(ro[5], rel=1e-6)
    assert mp.a.item() == pytest.approx(mo.a.item(), rel=1e-4)


def test_nogroup_single_sample_is_nan_like_the_reference():
    """The pinned `nogroup` fixture (tests/golden/trainer.json): one prompt, one sample, world size 1 -> the unbiased std of
    one reward is NaN, and so are the advantage, the loss and the gradient norm in the reference; the HIP path must not
    turn that into a number."""
    ro, rp, tro, trp, _, _ = _nogroup_case(1, [0.3])
    assert torch.isnan(tro["advantages"]).all() and torch.isnan(trp["advantages"]).all()
    assert rp[0] != rp[0] and ro[0] != ro[0]                       # NaN loss on both sides
    assert rp[5] == pytest.approx(ro[5], rel=1e-6)
