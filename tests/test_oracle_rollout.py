"""Pins oracle/rollout.py bit-exactly against the reference's run_sample_step outputs."""
from argparse import Namespace

import pytest
import torch

from helpers import assert_same, host_matches_fixture_host, load_golden
from oracle import rollout as R
from oracle import solver as S
from toy_model import ToyTransformer

T_, M_ = load_golden("rollout")


EXACT = host_matches_fixture_host()


def eq(a, b):
    assert_same(a, b, exact=EXACT)


@pytest.mark.parametrize("inject", [False, True])
@pytest.mark.parametrize("case", M_["cases"], ids=lambda c: c["tag"])
def test_rollout(case, inject):
    a = Namespace(**case["args"])
    # (inject: the noise the reference drew -- randn_tensor AND the DanceGRPO step's randn_like -- is replayed from the
    # fixture instead of being re-drawn from the seeded generator: pins the rollout independently of the RNG stream)
    T = case["T"]
    sig = S.sd3_time_shift(a.shift, torch.linspace(1, 0, T + 1))
    det = [True] * T
    for i in case["window"]:
        det[i] = False
    model = ToyTransformer(64, seed=3)
    tag = case["tag"]
    noises = [T_[f"{tag}/noise{k}"] for k in range(case["n_noise"])] if inject else None
    torch.manual_seed(4242)
    with torch.no_grad():
        z, lat, all_lat, all_lp = R.run_sample_step(a, T_["in/z0"], range(T), sig, model, T_["in/ehs"], T_["in/pooled"],
                                                    T_["in/text_ids"], T_["in/img_ids"], True, det, noises=noises)
    assert all_lp.shape[1] == case["steps_run"]
    eq(z, T_[f"{tag}/z"])
    eq(lat, T_[f"{tag}/latents"])
    eq(all_lat, T_[f"{tag}/all_latents"])
    eq(all_lp, T_[f"{tag}/all_log_probs"])


def test_flash_schedule_anchors():
    """SURVEY 8c anchors: T=25 shift 3: window [0,1] ratio .4 -> 10 steps; [10..13] -> 17; [21,22] ratio .2 -> 23."""
    sig = S.sd3_time_shift(3.0, torch.linspace(1, 0, 26))
    for window, ratio, steps in (([0, 1], 0.4, 10), ([10, 11, 12, 13], 0.4, 17), ([21, 22], 0.2, 23)):
        det = [i not in window for i in range(25)]
        s2, last = R.flash_schedule(sig, det, ratio, 3.0)
        assert s2.numel() - 1 == steps and last == window[-1]
    assert s2[-1] != 0  # single post step: the rollout stops at sigma != 0 (App. C-9)
