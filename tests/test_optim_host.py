"""Host-side LR schedules (mixgrpo_amd/optim.py `get_scheduler`) = the reference's `get_scheduler(args.lr_scheduler, ...)`
(fastvideo/train_grpo_flux.py:726-734; diffusers' optimization.py, absent here).  The pin that IS available: transformers'
same-named LambdaLR schedules (diffusers' file is a copy of them), stepped side by side on a torch optimizer."""
import pytest
import torch

from mixgrpo_amd.optim import SCHEDULER_NAMES, get_scheduler


class _Opt:
    def __init__(self, lr):
        self.param_groups = [{"lr": lr}]


def _theirs(name, opt, warmup, total, num_cycles, power):
    from transformers import optimization as TO
    if name == "constant":
        return TO.get_constant_schedule(opt)
    if name == "constant_with_warmup":
        return TO.get_constant_schedule_with_warmup(opt, warmup)
    if name == "linear":
        return TO.get_linear_schedule_with_warmup(opt, warmup, total)
    if name == "cosine":
        return TO.get_cosine_schedule_with_warmup(opt, warmup, total)                 # diffusers passes no num_cycles here
    if name == "cosine_with_restarts":
        return TO.get_cosine_with_hard_restarts_schedule_with_warmup(opt, warmup, total, num_cycles=num_cycles)
    return TO.get_polynomial_decay_schedule_with_warmup(opt, warmup, total, lr_end=1e-7, power=power)


@pytest.mark.parametrize("name", SCHEDULER_NAMES)
@pytest.mark.parametrize("warmup,total,num_cycles,power", [(0, 1000000, 1, 1.0), (5, 40, 3, 2.0), (3, 20, 1, 1.0)])
def test_schedules_follow_the_reference_formulas(name, warmup, total, num_cycles, power):
    lr = 1e-5
    p = torch.nn.Parameter(torch.zeros(1))
    topt = torch.optim.AdamW([p], lr=lr)
    theirs = _theirs(name, topt, warmup, total, num_cycles, power)
    mine = get_scheduler(name, _Opt(lr), num_warmup_steps=warmup, num_training_steps=total, num_cycles=num_cycles, power=power)
    for step in range(60):
        assert mine.get_last_lr()[0] == pytest.approx(theirs.get_last_lr()[0], rel=1e-12, abs=1e-20), (name, step)
        topt.step()
        theirs.step()
        mine.step()


def test_unknown_name_fails_with_the_list():
    with pytest.raises(ValueError, match="constant_with_warmup"):
        get_scheduler("piecewise_constant", _Opt(1e-5))
