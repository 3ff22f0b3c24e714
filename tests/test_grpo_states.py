"""Product window scheduler vs the index sequences recorded from the reference's GRPOTrainingStates."""
import json
import os

import pytest

from helpers import GOLDEN
from mixgrpo_amd.grpo_states import GRPOTrainingStates

CASES = json.load(open(os.path.join(GOLDEN, "windows.json")))


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c["params"].items())[:80])
def test_window_sequence(case):
    st = GRPOTrainingStates(**case["params"])
    for it, exp in enumerate(case["sequence"]):
        assert [int(t) for t in st.get_current_timesteps()] == exp, it
        st.update_iteration(seed=None if case["seeds"] is None else case["seeds"] + it)
    assert int(st.cur_timestep) == case["final"]["cur_timestep"]
    assert int(st.cur_iter_in_group) == case["final"]["cur_iter_in_group"]


def test_survey_anchor_sequence():
    st = GRPOTrainingStates(iters_per_group=2, group_size=2, max_timesteps=6, prog_overlap=True, prog_overlap_step=1,
                            roll_back=True)
    seq = []
    for _ in range(16):
        seq.append(st.get_current_timesteps())
        st.update_iteration()
    assert seq == [[0, 1]] * 2 + [[1, 2]] * 2 + [[2, 3]] * 2 + [[3, 4]] * 2 + [[4, 5]] * 2 + [[5]] * 2 + [[]] * 2 + [[0, 1]] * 2


def test_invalid_strategy_raises():
    st = GRPOTrainingStates(2, 2, 6, sample_strategy="bogus")
    with pytest.raises(ValueError):
        st.update_iteration()
    assert not GRPOTrainingStates(2, 2, 6, sample_strategy="random").is_training_complete()
