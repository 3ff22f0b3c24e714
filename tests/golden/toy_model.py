"""Deterministic toy velocity model used by the golden-fixture generator and by the tests.

It has the call signature of the FLUX transformer as the reference calls it
(/root/reference/fastvideo/utils/sampling_utils.py:68-82, train_grpo_flux.py:134-144)
and returns a one-tuple with a **bf16** tensor, because `torch.autocast("cuda")` is a
no-op on CPU and the solver's dtype behaviour depends on a bf16
model output.  This file is test infrastructure, not part of the product.
"""
import torch
from torch import nn


class ToyTransformer(nn.Module):
    def __init__(self, channels: int = 64, seed: int = 0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.w = nn.Parameter(torch.randn(channels, channels, generator=g) * 0.08)
        self.b = nn.Parameter(torch.randn(channels, generator=g) * 0.02)
        self.t = nn.Parameter(torch.randn(channels, generator=g) * 0.3)
        self.config = {"toy": True, "in_channels": channels}

    def forward(self, hidden_states, encoder_hidden_states, timestep, guidance, txt_ids,
                pooled_projections, img_ids, joint_attention_kwargs=None, return_dict=False):
        x = hidden_states.to(torch.float32)
        ctx = encoder_hidden_states.to(torch.float32).mean(dim=(1, 2)).view(-1, 1, 1)
        pool = pooled_projections.to(torch.float32).mean(dim=-1).view(-1, 1, 1)
        tt = timestep.to(torch.float32).view(-1, 1, 1)
        pos = (img_ids.to(torch.float32)[..., 1] * 0.01 + img_ids.to(torch.float32)[..., 2] * 0.003)
        pos = pos.reshape(1, -1, 1)
        h = x @ self.w + self.b + tt * self.t + 0.1 * ctx + 0.05 * pool + pos
        h = h * (guidance.to(torch.float32).view(-1, 1, 1) / 3.5)
        return (h.to(torch.bfloat16),)

    def clip_grad_norm_(self, max_norm):
        return torch.nn.utils.clip_grad_norm_(self.parameters(), max_norm)


class ElementwiseToy(nn.Module):
    """Velocity model made of separately-rounded fp32 elementwise ops only, so CPU and GPU agree bit for bit."""

    def __init__(self):
        super().__init__()
        self.a = nn.Parameter(torch.tensor(0.37))
        self.config = {"toy": True}

    def forward(self, hidden_states, encoder_hidden_states, timestep, guidance, txt_ids,
                pooled_projections, img_ids, joint_attention_kwargs=None, return_dict=False):
        x = hidden_states.to(torch.float32)
        tt = timestep.to(torch.float32).view(-1, 1, 1)
        pos = (img_ids.to(torch.float32)[..., 1] * 0.125 + img_ids.to(torch.float32)[..., 2] * 0.0625).reshape(1, -1, 1)
        h = x * self.a
        h = h + tt
        h = h - pos
        return (h.to(torch.bfloat16),)

    def clip_grad_norm_(self, max_norm):
        return torch.nn.utils.clip_grad_norm_(self.parameters(), max_norm)
