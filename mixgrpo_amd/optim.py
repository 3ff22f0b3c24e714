"""Fused AdamW + constant-with-warmup schedule on the flat parameter store.

Drop-in for what the reference builds at fastvideo/train_grpo_flux.py:715-734 (torch.optim.AdamW with
betas (0.9, 0.999), eps 1e-8, and diffusers' get_scheduler("constant_with_warmup")), running as ONE HIP kernel
over the flat fp32 master weights / grads / moments that also refreshes the bf16 compute copy, with the
clip-by-global-norm factor (train_grpo_flux.py:606) applied inside the same pass.
"""
import math

import torch

from . import ops


class FusedAdamW:
    def __init__(self, model, lr=1e-5, betas=(0.9, 0.999), weight_decay=1e-4, eps=1e-8):
        self.model = model
        self.store = model.store
        self.lr = float(lr)
        self.base_lr = float(lr)
        self.betas = betas
        self.weight_decay = float(weight_decay)
        self.eps = float(eps)
        self.step_count = 0
        dev = self.store.device
        self.m = torch.zeros(self.store.numel, dtype=torch.float32, device=dev)
        self.v = torch.zeros(self.store.numel, dtype=torch.float32, device=dev)
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.param_groups = [{"lr": self.lr}]     # what lr schedulers / loggers poke at

    def zero_grad(self, set_to_none=False):
        g = self.store.g32
        if g is not None:
            g.zero_()

    def grad_sqnorm(self):
        """sum g^2 of the local gradient buffer into self.gnorm_sq (device, no sync)."""
        ops.sqnorm(self.store.ensure_grad(), self.gnorm_sq)
        return self.gnorm_sq

    def step(self, max_grad_norm=None, grad_scale=1.0, gnorm_sq=None):
        """AdamW step.  With `max_grad_norm`, gradients are scaled by min(1, max_norm / (|grad_scale * g| + 1e-6))
        inside the kernel; `gnorm_sq` (device scalar, sum of UNSCALED g^2) defaults to the local buffer's."""
        self.step_count += 1
        lr = self.param_groups[0]["lr"]
        nsq = None
        if max_grad_norm is not None:
            nsq = gnorm_sq if gnorm_sq is not None else self.grad_sqnorm()
        ops.adamw_step(self.store.w32, self.store.w16, self.store.ensure_grad(), self.m, self.v, lr, self.betas[0],
                       self.betas[1], self.eps, self.weight_decay, self.step_count, nsq,
                       0.0 if max_grad_norm is None else float(max_grad_norm), grad_scale)

    def state_dict(self):
        return {"step": self.step_count, "m": self.m, "v": self.v, "lr": self.param_groups[0]["lr"]}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.param_groups[0]["lr"] = float(sd["lr"])


SCHEDULER_NAMES = ("linear", "cosine", "cosine_with_restarts", "polynomial", "constant", "constant_with_warmup")


class LambdaSchedule:
    """diffusers `get_scheduler(name, optimizer, num_warmup_steps, num_training_steps, num_cycles, power)` as the reference
    builds it (fastvideo/train_grpo_flux.py:726-734: `num_training_steps=1000000`, `num_cycles=args.lr_num_cycles`,
    `power=args.lr_power`): lr = base_lr * factor(step), stepped once per optimizer step.  The factors are diffusers
    `optimization.py`'s LambdaLR lambdas (the same formulas as transformers.optimization, against which
    tests/test_optim_host.py holds them step by step; diffusers itself is absent: parity with it is unpinned).  As in
    diffusers, `num_cycles` reaches only "cosine_with_restarts" ("cosine" runs its default half cycle) and `power` only
    "polynomial" (lr_end 1e-7); "piecewise_constant" needs `step_rules`, which the reference never passes."""

    def __init__(self, optimizer, name="constant", num_warmup_steps=0, num_training_steps=1000000, num_cycles=1, power=1.0):
        if name not in SCHEDULER_NAMES:
            raise ValueError(f"lr_scheduler {name!r} is not supported; choose one of {', '.join(SCHEDULER_NAMES)} "
                             "(piecewise_constant needs step_rules, which the reference's get_scheduler call never passes)")
        self.opt = optimizer
        self.name = name
        self.warmup = int(num_warmup_steps) if name != "constant" else 0
        self.total = int(num_training_steps)
        self.num_cycles = num_cycles
        self.power = power
        self.base = [g["lr"] for g in optimizer.param_groups]
        self.n = 0
        self._apply()

    def _factor(self, lr_init=None):
        n, w, T = self.n, self.warmup, self.total
        if self.name == "constant":
            return 1.0
        if n < w:
            return float(n) / float(max(1.0 if self.name == "constant_with_warmup" else 1, w))
        if self.name == "constant_with_warmup":
            return 1.0
        if self.name == "linear":
            return max(0.0, float(T - n) / float(max(1, T - w)))
        progress = float(n - w) / float(max(1, T - w))
        if self.name == "cosine":
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * 0.5 * 2.0 * progress)))
        if self.name == "cosine_with_restarts":
            if progress >= 1.0:
                return 0.0
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((float(self.num_cycles) * progress) % 1.0))))
        # polynomial
        lr_end = 1e-7
        if n > T:
            return lr_end / lr_init
        pct_remaining = 1 - (n - w) / (T - w)
        return ((lr_init - lr_end) * pct_remaining ** self.power + lr_end) / lr_init

    def _apply(self):
        for g, b in zip(self.opt.param_groups, self.base):
            g["lr"] = b * self._factor(b)

    def step(self):
        self.n += 1
        self._apply()

    def get_last_lr(self):
        return [g["lr"] for g in self.opt.param_groups]


def get_scheduler(name, optimizer, num_warmup_steps=0, num_training_steps=1000000, num_cycles=1, power=1.0, last_epoch=-1):
    """The reference's `get_scheduler(...)` call (train_grpo_flux.py:726-734); `last_epoch` is accepted for the signature
    (the reference always passes -1: its resume is unimplemented; ours restores `n` from the resume state)."""
    return LambdaSchedule(optimizer, name, num_warmup_steps, num_training_steps, num_cycles, power)


class ConstantWithWarmup(LambdaSchedule):
    """diffusers get_scheduler("constant_with_warmup"): lr * min(1, step / max(1, warmup))."""

    def __init__(self, optimizer, num_warmup_steps=0, last_epoch=-1):
        super().__init__(optimizer, "constant_with_warmup", num_warmup_steps)
