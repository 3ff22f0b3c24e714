"""Fused AdamW + constant-with-warmup schedule on the flat parameter store.

Drop-in for what the reference builds at fastvideo/train_grpo_flux.py:715-734 (torch.optim.AdamW with
betas (0.9, 0.999), eps 1e-8, and diffusers' get_scheduler("constant_with_warmup")), running as ONE HIP kernel
over the flat fp32 master weights / grads / moments that also refreshes the bf16 compute copy, with the
clip-by-global-norm factor (train_grpo_flux.py:606) applied inside the same pass.
"""
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


class ConstantWithWarmup:
    """diffusers get_scheduler("constant_with_warmup"): lr * min(1, step / max(1, warmup))."""

    def __init__(self, optimizer, num_warmup_steps=0, last_epoch=-1):
        self.opt = optimizer
        self.warmup = int(num_warmup_steps)
        self.base = [g["lr"] for g in optimizer.param_groups]
        self.n = 0
        self._apply()

    def _factor(self):
        if self.n < self.warmup:
            return float(self.n) / float(max(1.0, self.warmup))
        return 1.0

    def _apply(self):
        for g, b in zip(self.opt.param_groups, self.base):
            g["lr"] = b * self._factor()

    def step(self):
        self.n += 1
        self._apply()

    def get_last_lr(self):
        return [g["lr"] for g in self.opt.param_groups]
