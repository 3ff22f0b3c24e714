"""CPU restatement of the reference's mixed-model inference loop (fastvideo/sample/sample_flux.py:249-264, 308-365)
on the oracle MMDiT -- TEST INFRASTRUCTURE ONLY (imported by tests/ alone).

PINNED to the reference (tests/test_sampler_host.py, fixtures tests/golden/sampler_schedule.json produced by
gen_fixtures.py from the reference's OWN vendored helpers, fastvideo/models/flux_hf/pipeline_flux.py:73-84,87-145):
`calculate_shift` bit for bit, and the unshifted sigma grid + mu that `retrieve_timesteps` hands to the scheduler.
PARITY UNPINNED: what the scheduler does with them -- FlowMatchEulerDiscreteScheduler.set_timesteps (the dynamic shift
sigma' = e^mu / (e^mu + 1/sigma - 1)) and .step (Euler update in fp32) live in diffusers==0.32.2, absent offline, and the
reference holds no fixture of either; this follows their published algorithm.
"""
import math

import numpy as np

import torch

from . import mmdit as OM


def calculate_shift(image_seq_len, base_seq_len=256, max_seq_len=4096, base_shift=0.5, max_shift=1.15):
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    return image_seq_len * m + (base_shift - m * base_seq_len)


def sigma_grid(num_inference_steps):
    """np.linspace(1.0, 1 / T, T) of sample_flux.py:249, in float64 (numpy's own: torch.linspace differs in the last bits)."""
    return torch.from_numpy(np.linspace(1.0, 1 / num_inference_steps, num_inference_steps))


def sigmas_for(num_inference_steps, n_img):
    s = sigma_grid(num_inference_steps)
    mu = calculate_shift(n_img)
    s = (math.exp(mu) / (math.exp(mu) + (1.0 / s - 1.0))).float()
    return torch.cat([s, torch.zeros(1)])


def dual_sample(P_base, P_new, cfg, latents, prompt_embeds, pooled, text_ids, img_ids, num_inference_steps,
                mix_sampling_steps, guidance_scale=3.5):
    """latents [B, N, 64] bf16 -> packed latents after the mixed loop (reference :308-365, no true-CFG)."""
    sig = sigmas_for(num_inference_steps, latents.shape[1])
    B = latents.shape[0]
    g = torch.full([B], guidance_scale)
    for i in range(num_inference_steps):
        t = (sig[i] * 1000.0).expand(B).to(latents.dtype)
        P = P_new if i < mix_sampling_steps else P_base
        v = OM.forward(P, cfg, latents.float(), prompt_embeds.float(), (t / 1000).float(), g, text_ids, pooled.float(),
                       img_ids).to(torch.bfloat16)
        dt = (sig[i + 1] - sig[i]).item()
        latents = (latents.float() + dt * v.float()).to(v.dtype)
    return latents
