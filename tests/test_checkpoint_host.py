"""Host side of the resume state (mixgrpo_amd/checkpoint.py): every window-scheduler strategy survives a
save -> load round trip (the `random` strategy keeps a numpy integer in the reference, grpo_states.py:101-102; JSON must not
choke on it), and a failed save leaves no half-written resume directory."""
import json
import os

import pytest
import torch

from mixgrpo_amd.checkpoint import load_resume_state, save_resume_state
from mixgrpo_amd.grpo_states import GRPOTrainingStates
from mixgrpo_amd.optim import ConstantWithWarmup


class _Opt:
    """What save/load_resume_state touch of FusedAdamW, on CPU tensors."""

    def __init__(self, n=32, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.m, self.v = torch.randn(n, generator=g), torch.rand(n, generator=g)
        self.step_count, self.base_lr = 7, 1e-5
        self.param_groups = [{"lr": 1e-5}]

    def state_dict(self):
        return {"step": self.step_count, "m": self.m, "v": self.v, "lr": self.param_groups[0]["lr"]}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.param_groups[0]["lr"] = float(sd["lr"])


@pytest.mark.parametrize("strategy,kw", [
    ("progressive", dict(prog_overlap=True, prog_overlap_step=1, roll_back=True)),
    ("random", dict()),
    ("decay", dict(max_iters_per_group=6, min_iters_per_group=2)),
    ("exp_decay", dict(exp_decay_thre_timestep=2, exp_decay_k=0.3, prog_overlap=True)),
])
def test_resume_roundtrip_every_strategy(tmp_path, strategy, kw):
    st = GRPOTrainingStates(iters_per_group=3, group_size=2, max_timesteps=10, sample_strategy=strategy, **kw)
    for i in range(11):
        st.update_iteration(seed=100 + i)
    opt = _Opt()
    sched = ConstantWithWarmup(opt, 4)
    for _ in range(3):
        sched.step()
    save_resume_state(str(tmp_path), opt, sched, st, global_step=11)
    assert sorted(os.listdir(tmp_path)) == ["optimizer.safetensors", "trainer_state.json"]       # no temporaries left
    raw = json.load(open(tmp_path / "trainer_state.json"))
    assert isinstance(raw["grpo_states"]["cur_timestep"], int)
    st2 = GRPOTrainingStates(iters_per_group=3, group_size=2, max_timesteps=10, sample_strategy=strategy, **kw)
    opt2 = _Opt(seed=1)
    sched2 = ConstantWithWarmup(opt2, 4)
    assert load_resume_state(str(tmp_path), opt2, sched2, st2) == 11
    assert torch.equal(opt2.m, opt.m) and torch.equal(opt2.v, opt.v) and opt2.step_count == 7
    assert sched2.n == sched.n and sched2.get_last_lr() == sched.get_last_lr()
    assert st2.get_current_timesteps() == st.get_current_timesteps()
    for i in range(9):                                       # and the two schedulers stay in step afterwards
        st.update_iteration(seed=500 + i)
        st2.update_iteration(seed=500 + i)
        assert st2.get_current_timesteps() == st.get_current_timesteps()
        assert st2.cur_iter_in_group == st.cur_iter_in_group


def test_failed_save_leaves_no_partial_state(tmp_path):
    st = GRPOTrainingStates(iters_per_group=3, group_size=2, max_timesteps=10)
    st.exp_decay_k = object()                                 # something JSON cannot take
    with pytest.raises(TypeError):
        save_resume_state(str(tmp_path), _Opt(), None, st, global_step=1)
    assert os.listdir(tmp_path) == []


def test_resume_position_records_the_epoch(tmp_path):
    """`checkpoint-{step}-{epoch}` with epoch > 0 is the normal case under the shipped launcher (max_train_steps per epoch,
    a checkpoint every checkpointing_steps): the resume state carries the epoch, the position inside it and the number of
    train steps done over all epochs; a state written before the epoch was recorded reads as epoch 0."""
    from mixgrpo_amd.checkpoint import load_resume_position
    save_resume_state(str(tmp_path), _Opt(), None, None, global_step=49, epoch=2, steps_done=2 * 300 + 49)
    assert load_resume_position(str(tmp_path)) == (2, 49, 649)
    assert load_resume_state(str(tmp_path), _Opt(seed=1)) == 49
    raw = json.load(open(tmp_path / "trainer_state.json"))
    del raw["epoch"], raw["steps_done"]
    with open(tmp_path / "trainer_state.json", "w") as f:
        json.dump(raw, f)
    assert load_resume_position(str(tmp_path)) == (0, 49, 49)
