"""Mixed-model inference sampler: the reference's `DualFluxPipeline` (fastvideo/sample/sample_flux.py:25-401) on the
HIP MMDiT.

The first `mix_sampling_steps` denoising steps run the GRPO-tuned transformer (`transformer_new`, reference :319-330),
the rest the base transformer (:331-342); the schedule is the FLUX pipeline's dynamically shifted flow-matching
schedule (`calculate_shift` mu from the image token count, :249-264, then diffusers'
FlowMatchEulerDiscreteScheduler with `use_dynamic_shifting`), the update an Euler step in fp32 cast back to the latent
dtype.  Text encoders are outside the hot path: the sampler takes cached prompt embeddings and returns packed latents, or -- given
a VAE (`mixgrpo_amd.vae.AutoencoderKL`, the HIP decode) -- images, through the reference's tail (:387-393).

`calculate_shift` and the sigma grid handed to the scheduler are held bit for bit to the reference's vendored helpers
(fastvideo/models/flux_hf/pipeline_flux.py:73-84,87-145; tests/test_sampler_host.py, tests/golden/sampler_schedule.json).
The scheduler itself (dynamic shift, Euler step) lives in diffusers==0.32.2 (absent offline): that arithmetic restates the
published algorithm -- PARITY UNPINNED against diffusers, checked against oracle/sampler.py (the same restatement on the
CPU oracle MMDiT).
"""
import math

import numpy as np
import torch

from .latents import prepare_latent_image_ids


def calculate_shift(image_seq_len, base_seq_len=256, max_seq_len=4096, base_shift=0.5, max_shift=1.15):
    """mu = linear interpolation of the shift in the image token count (diffusers pipeline_flux.calculate_shift)."""
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    b = base_shift - m * base_seq_len
    return image_seq_len * m + b


def flow_match_sigmas(num_inference_steps, mu, sigmas=None):
    """FlowMatchEulerDiscreteScheduler.set_timesteps(sigmas=linspace(1, 1/T, T), mu=mu) with dynamic shifting:
    sigma' = e^mu / (e^mu + (1/sigma - 1)); returns (sigmas' + [0], timesteps = sigmas' * 1000) as fp32 tensors."""
    if sigmas is None:
        # the reference's grid is numpy's (sample_flux.py:249); torch.linspace differs from it in the last bits
        sigmas = torch.from_numpy(np.linspace(1.0, 1 / num_inference_steps, num_inference_steps))
    else:
        sigmas = torch.as_tensor(sigmas, dtype=torch.float64)
    shifted = math.exp(mu) / (math.exp(mu) + (1.0 / sigmas - 1.0))
    shifted = shifted.to(torch.float32)
    return torch.cat([shifted, torch.zeros(1)]), shifted * 1000.0


class DualFluxSampler:
    """`transformer`: base model, `transformer_new`: tuned model (both mixgrpo_amd.flux.FluxTransformer2DModel)."""

    def __init__(self, transformer, transformer_new=None, vae=None):
        self.transformer = transformer
        self.transformer_new = transformer_new
        self.vae = vae            # mixgrpo_amd.vae.AutoencoderKL (or any object with .config / .decode): needed for image outputs

    def load_new_model(self, model_path):
        """Reference :27-33: a second transformer with the tuned weights (a safetensors file in diffusers key names)."""
        from safetensors.torch import load_file

        from .flux import FluxTransformer2DModel
        self.transformer_new = FluxTransformer2DModel(self.transformer.cfg, device=self.transformer.store.device)
        self.transformer_new.load_state_dict(load_file(model_path), strict=True)

    @torch.no_grad()
    def __call__(self, prompt_embeds, pooled_prompt_embeds, height=1024, width=1024, num_inference_steps=28,
                 mix_sampling_steps=10, sigmas=None, guidance_scale=3.5, generator=None, latents=None,
                 true_cfg_scale=1.0, negative_prompt_embeds=None, negative_pooled_prompt_embeds=None,
                 text_ids=None, max_sequence_length=512, output_type="latent"):
        """`output_type`: "latent" (packed latents, the default here) | "pt" ([B, 3, H, W] in [0, 1]) | "np" ([B, H, W, 3] float32)
        | "pil" -- the reference's tail (:387-393): unpack, `/ scaling_factor + shift_factor`, `vae.decode`, postprocess."""
        dev = self.transformer.store.device
        if output_type != "latent" and self.vae is None:
            raise ValueError(f'output_type="{output_type}" needs a VAE: DualFluxSampler(..., vae=AutoencoderKL.from_pretrained(...))')
        if mix_sampling_steps > 0 and self.transformer_new is None:
            raise ValueError("mix_sampling_steps > 0 needs the tuned model: call load_new_model() first")
        B = prompt_embeds.shape[0]
        prompt_embeds = prompt_embeds.to(dev)
        pooled_prompt_embeds = pooled_prompt_embeds.to(dev)
        dtype = prompt_embeds.dtype
        hl, wl = 2 * (int(height) // 16), 2 * (int(width) // 16)        # latent grid (VAE factor 8, even)
        C = self.transformer.cfg.in_channels // 4
        if latents is None:
            z = torch.randn(B, C, hl, wl, generator=generator, device=dev if generator is None else generator.device,
                            dtype=dtype).to(dev)
            latents = z.view(B, C, hl // 2, 2, wl // 2, 2).permute(0, 2, 4, 1, 3, 5).reshape(B, (hl // 2) * (wl // 2), C * 4)
        latents = latents.to(dev).contiguous()
        img_ids = prepare_latent_image_ids(B, hl // 2, wl // 2, dev, dtype)
        if text_ids is None:
            text_ids = torch.zeros(prompt_embeds.shape[1], 3, device=dev, dtype=dtype)
        sig, timesteps = flow_match_sigmas(num_inference_steps, calculate_shift(latents.shape[1]), sigmas)
        guidance = None
        if self.transformer.cfg.guidance_embeds:
            guidance = torch.full([1], guidance_scale, device=dev, dtype=torch.float32).expand(B)
        do_true_cfg = true_cfg_scale > 1 and negative_prompt_embeds is not None and negative_pooled_prompt_embeds is not None
        for i in range(len(timesteps)):
            t = timesteps[i].to(dev).expand(B).to(latents.dtype)
            model = self.transformer_new if i < mix_sampling_steps else self.transformer
            kw = dict(hidden_states=latents, timestep=t / 1000, guidance=guidance, txt_ids=text_ids, img_ids=img_ids,
                      joint_attention_kwargs=None, return_dict=False)
            noise_pred = model(pooled_projections=pooled_prompt_embeds, encoder_hidden_states=prompt_embeds, **kw)[0]
            if do_true_cfg:                                              # reference :344-359 (always the base model)
                neg = self.transformer(pooled_projections=negative_pooled_prompt_embeds.to(dev),
                                       encoder_hidden_states=negative_prompt_embeds.to(dev), **kw)[0]
                noise_pred = neg + true_cfg_scale * (noise_pred - neg)
            # FlowMatchEulerDiscreteScheduler.step: fp32 Euler update, cast back to the model-output dtype
            dt = (sig[i + 1] - sig[i]).item()
            latents = (latents.to(torch.float32) + dt * noise_pred.to(torch.float32)).to(noise_pred.dtype)
        if output_type == "latent":
            return latents
        return self.decode_latents(latents, height, width, output_type)

    def decode_latents(self, latents, height, width, output_type="pt"):
        """Reference :390-393 with diffusers' VaeImageProcessor.postprocess restated for its three plain outputs:
        denormalise `(x / 2 + 0.5).clamp(0, 1)` IN THE DECODER'S DTYPE (bf16), then `.float()`; "np" = NHWC float32;
        "pil" = uint8 `(x * 255).round()` images.  Parity unpinned: diffusers is not importable here."""
        from .latents import unpack_latents
        cfg = self.vae.config
        z = unpack_latents(latents, height, width, 8)
        z = (z / cfg.scaling_factor) + cfg.shift_factor
        image = self.vae.decode(z, return_dict=False)[0]
        image = (image / 2 + 0.5).clamp(0, 1).float()
        if output_type == "pt":
            return image
        arr = image.permute(0, 2, 3, 1).cpu().numpy()
        if output_type == "np":
            return arr
        if output_type == "pil":
            from PIL import Image
            return [Image.fromarray((a * 255).round().astype("uint8")) for a in arr]
        raise ValueError(f"unknown output_type {output_type}")
