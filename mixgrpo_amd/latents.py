"""Latent layout helpers: drop-in for the reference's `prepare_latent_image_ids`, `pack_latents`,
`unpack_latents` (fastvideo/train_grpo_flux.py:80-115), on HIP (csrc/solver.hip pack kernels)."""
import torch

from ._lib import check, lib, ptr, stream


def prepare_latent_image_ids(batch_size, height, width, device, dtype):
    """ids[h*w, 3] = (0, row, col) -- tiny host-built table (train_grpo_flux.py:80-91)."""
    ids = torch.zeros(height, width, 3)
    ids[..., 1] += torch.arange(height)[:, None]
    ids[..., 2] += torch.arange(width)[None, :]
    return ids.reshape(height * width, 3).to(device=device, dtype=dtype)


def pack_latents(latents, batch_size, num_channels_latents, height, width):
    """[B,C,H,W] -> [B,(H/2)(W/2),4C] (train_grpo_flux.py:94-99)."""
    x = latents.contiguous()
    out = torch.empty(batch_size, (height // 2) * (width // 2), num_channels_latents * 4, dtype=x.dtype,
                      device=x.device)
    check(lib().mgx_pack_latents(ptr(x), ptr(out), batch_size, num_channels_latents, height, width, x.element_size(),
                                 stream()))
    return out


def unpack_latents(latents, height, width, vae_scale_factor):
    """[B,(H/2)(W/2),4C] -> [B,C,H,W]; height/width are IMAGE sizes (train_grpo_flux.py:102-115)."""
    b, n, ch = latents.shape
    h = 2 * (int(height) // (vae_scale_factor * 2))
    w = 2 * (int(width) // (vae_scale_factor * 2))
    x = latents.contiguous()
    out = torch.empty(b, ch // 4, h, w, dtype=x.dtype, device=x.device)
    check(lib().mgx_unpack_latents(ptr(x), ptr(out), b, ch // 4, h, w, x.element_size(), stream()))
    return out
