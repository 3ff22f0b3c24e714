"""Decode + reward stage next to the hot path: builds the `reward_function(latents, captions)` the
engine's `sample_reference_model` / `train_one_step` call out of a VAE and the reference's reward models.

What the reference does per sample after a rollout (fastvideo/train_grpo_flux.py:279-316):
    latents = unpack_latents(latents, h, w, 8);  latents = latents / 0.3611 + 0.1159          (FLUX VAE scaling / shift)
    image   = vae.decode(latents, return_dict=False)[0];  pil = VaeImageProcessor(16).postprocess(image)
    rewards, successes, rewards_dict, successes_dict = compute_reward([pil], [caption], reward_function, reward_weights)
and then either sums pre-weighted rewards (`reward_aggr`) or keeps per-model lists keyed by the reward CLASS NAME
(`advantage_aggr`, :305-316,467).  The fork's own `compute_reward` is broken (reward_model/utils.py:4-15 returns an
undefined name), so the contract here follows the call sites.  The VAE and the reward networks themselves are third-party
weights that are not available offline: they are passed in as callables.
"""
import torch

from .latents import unpack_latents

FLUX_VAE_SCALING = 0.3611
FLUX_VAE_SHIFT = 0.1159


def decode_latents(vae, latents, height, width, image_processor=None, vae_scale_factor=8):
    """Packed latents [n, N, 64] -> decoded images (reference :284-289): unpack on the GPU, undo the FLUX latent
    normalisation, `vae.decode(..., return_dict=False)[0]`, optional `image_processor.postprocess`."""
    with torch.inference_mode(), torch.autocast("cuda", dtype=torch.bfloat16):
        z = unpack_latents(latents, height, width, vae_scale_factor)
        z = (z / FLUX_VAE_SCALING) + FLUX_VAE_SHIFT
        if hasattr(vae, "enable_tiling"):
            vae.enable_tiling()
        image = vae.decode(z, return_dict=False)[0]
        return image_processor.postprocess(image) if image_processor is not None else image


def compute_reward(images, prompts, reward_models, reward_weights):
    """-> (rewards, successes, rewards_dict, successes_dict) as the trainer's call site consumes them (:299-316):
    `rewards[i] = sum_m w_m * r_m[i]`, `rewards_dict[ClassName] = [r_m[i]]`, successes all 1."""
    assert len(images) == len(prompts), \
        f"length of `images` ({len(images)}) must be equal to length of `input_prompts` ({len(prompts)})"
    rewards_dict, successes_dict = {}, {}
    total = [0.0] * len(images)
    for name, model in reward_models.items():
        r = [float(v) for v in model(images, prompts)]
        assert len(r) == len(images), f"reward model {name} returned {len(r)} scores for {len(images)} images"
        rewards_dict[name] = r
        successes_dict[name] = [1] * len(r)
        w = float(reward_weights.get(name, 1.0))
        total = [t + w * v for t, v in zip(total, r)]
    return total, [1] * len(images), rewards_dict, successes_dict


def make_reward_function(vae, reward_models, reward_weights, height, width, image_processor=None):
    """The engine-side contract: `reward_function(latents [n, N, 64], captions) -> (total [n], {ClassName: [n]})`."""
    def reward_function(latents, captions):
        images = decode_latents(vae, latents, height, width, image_processor)
        if torch.is_tensor(images):
            images = list(images)
        total, _, per_model, _ = compute_reward(images, list(captions), reward_models, reward_weights)
        return total, per_model
    return reward_function
