"""Sliding SDE-window scheduler (pure host integer logic).

Drop-in for `fastvideo.utils.grpo_states.GRPOTrainingStates` (reference fastvideo/utils/grpo_states.py:6-159):
same constructor fields, `get_current_timesteps()`, `update_iteration(seed)`, `roll_back_start()`,
`is_training_complete()`, `set_params()`.  Index-exact against tests/golden/windows.json, including the
shrinking and empty windows near `max_timesteps` before a roll-back.
"""
import math
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

_STRATEGIES = ("progressive", "random", "decay", "exp_decay")


@dataclass
class GRPOTrainingStates:
    iters_per_group: int
    group_size: int
    max_timesteps: int
    cur_timestep: int = 0
    cur_iter_in_group: int = 0
    sample_strategy: str = "progressive"
    prog_overlap: bool = False
    prog_overlap_step: int = 1
    max_iters_per_group: Optional[int] = None
    min_iters_per_group: Optional[int] = None
    roll_back: bool = False
    exp_decay_thre_timestep: int = 13
    exp_decay_k: float = 0.1

    def __post_init__(self):
        if self.sample_strategy == "decay":
            if self.max_iters_per_group is None:
                self.max_iters_per_group = self.iters_per_group
            if self.min_iters_per_group is None:
                self.min_iters_per_group = max(1, self.iters_per_group // 4)
        self.init_timestep = self.cur_timestep

    def set_params(self, params: dict):
        for k, v in params.items():
            setattr(self, k, v)

    # -- how many iterations the window stays where it is -------------------------------------------------
    def get_dynamic_iters_per_group(self) -> int:
        """Linear decay of the per-window budget from max_ to min_iters_per_group (strategy "decay")."""
        if self.sample_strategy != "decay":
            return self.iters_per_group
        frac = self.cur_timestep / self.max_timesteps
        budget = int(self.max_iters_per_group * (1 - frac) + self.min_iters_per_group * frac)
        return max(self.min_iters_per_group, budget)

    def get_exp_decay_iters_per_group(self):
        """ceil(iters_per_group * exp(-k * relu(t - threshold))) (strategy "exp_decay")."""
        if self.sample_strategy != "exp_decay":
            return self.iters_per_group
        past = max(0, self.cur_timestep - self.exp_decay_thre_timestep)
        return np.ceil(self.iters_per_group * np.exp(-self.exp_decay_k * past))

    def _window_budget(self):
        if self.sample_strategy == "decay":
            return self.get_dynamic_iters_per_group()
        if self.sample_strategy == "exp_decay":
            return self.get_exp_decay_iters_per_group()
        return self.iters_per_group

    # -- state transitions --------------------------------------------------------------------------------
    def update_iteration(self, seed=None) -> None:
        if self.sample_strategy not in _STRATEGIES:
            raise ValueError(f"Invalid sample strategy: {self.sample_strategy}")
        if self.sample_strategy == "random":
            # (a python int: the reference keeps numpy's int64 here, which its own range() accepts but JSON does not)
            self.cur_timestep = int(np.random.default_rng(seed).integers(0, self.max_timesteps - self.group_size + 1))
            return
        self.cur_iter_in_group += 1
        if self.cur_iter_in_group >= self._window_budget():
            self.cur_iter_in_group = 0
            self.cur_timestep += self.prog_overlap_step if self.prog_overlap else self.group_size
        if self.cur_timestep > self.max_timesteps:
            if self.roll_back:
                self.roll_back_start()
            else:
                self.cur_timestep = self.max_timesteps

    def roll_back_start(self) -> None:
        self.cur_timestep = self.init_timestep
        self.cur_iter_in_group = 0

    def get_current_timesteps(self) -> List[int]:
        """Solver-step indices of the current SDE window: [cur, min(cur+group_size, max_timesteps))."""
        return list(range(self.cur_timestep, min(self.cur_timestep + self.group_size, self.max_timesteps)))

    def is_training_complete(self) -> bool:
        return self.sample_strategy in ("progressive", "decay") and self.cur_timestep >= self.max_timesteps
