"""Pins oracle/trainer.py against the reference's train_one_step run end-to-end on CPU (toy model)."""
import math
import random
from argparse import Namespace

import pytest
import torch

from helpers import assert_same, host_matches_fixture_host, load_golden
from oracle import trainer as OT
from toy_model import ToyTransformer

T_, M_ = load_golden("trainer")
EXACT = host_matches_fixture_host()
TOL = 0.0 if EXACT else 1e-3


class _Sched:
    def step(self):
        pass


def run_case(case, train_fn=OT.train_one_step, model=None):
    a = Namespace(**case["args"])
    heads = list(case["rewards"].keys())
    w = case["weights"]

    def reward_fn(i, latents):
        rd = {h: [float(case["rewards"][h][i])] for h in heads}
        return [sum(float(w[h]) * rd[h][0] for h in heads)], rd

    model = model or ToyTransformer(64, seed=11)
    opt = torch.optim.AdamW(model.parameters(), lr=case["opt_lr"], betas=(0.9, 0.999), weight_decay=1e-4, eps=1e-8)
    tag = case["tag"]
    batch = (T_[f"{tag}/ehs"], T_[f"{tag}/pooled"], torch.zeros(1, 3), ["a toy prompt"])
    torch.manual_seed(714)
    random.seed(714)
    trace = {}
    res = train_fn(a, model, opt, _Sched(), batch, reward_fn, w, case["window"], 1.0, trace=trace)
    return res, model, trace


def close(a, b, tol=0.0):
    if isinstance(b, float) and math.isnan(b):
        return isinstance(a, float) and math.isnan(a)
    return a == b if tol == 0.0 else abs(a - b) <= tol * max(1.0, abs(b))


@pytest.mark.parametrize("case", M_["cases"], ids=lambda c: c["tag"])
def test_train_one_step(case):
    res, model, _ = run_case(case)
    ret = case["ret"]
    assert close(res[0], ret["total_loss"], TOL), (res[0], ret["total_loss"])
    assert close(res[1], ret["grad_norm"], TOL), (res[1], ret["grad_norm"])
    assert close(res[2], ret["policy_total_loss"], TOL)
    assert close(res[3], ret["kl_total_loss"], TOL)
    assert close(res[4], ret["total_clip_frac"], TOL)
    assert res[5] == ret["reward_mean"]
    for n, p in model.named_parameters():
        exp = T_[f"{case['tag']}/param_after/{n}"]
        assert_same(p.detach(), exp, exact=EXACT, rtol=1e-3, atol=1e-5)


def test_known_answer_advantages():
    """SURVEY 8c anchor: rewards .1,.2,.3,.4 -> advantages -1.16189, -0.38730, +0.38730, +1.16189."""
    adv = OT.group_advantages(torch.tensor([0.1, 0.2, 0.3, 0.4]), 4, 0.0)
    assert torch.allclose(adv, torch.tensor([-1.16189, -0.38730, 0.38730, 1.16189]), atol=1e-5)
    assert torch.equal(OT.group_advantages(torch.full((4,), 0.5), 4, 0.0), torch.zeros(4))  # App. C-12


def test_pack_unpack_roundtrip():
    x = torch.randn(2, 16, 8, 12)
    p = OT.pack_latents(x)
    assert p.shape == (2, 24, 64)
    assert torch.equal(OT.unpack_latents(p, 64, 96), x)
    ids = OT.prepare_latent_image_ids(4, 6, torch.float32)
    assert ids.shape == (24, 3) and ids[7].tolist() == [0.0, 1.0, 1.0]
