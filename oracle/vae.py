"""CPU oracle of the FLUX VAE decode -- TEST INFRASTRUCTURE ONLY (imported by tests/, never by the product path).

What it restates: `vae.enable_tiling(); image = vae.decode(latents, return_dict=False)[0]` of the reference's rollout
(fastvideo/train_grpo_flux.py:279-289) with `vae = AutoencoderKL.from_pretrained(..., subfolder="vae", torch_dtype=torch.bfloat16)`
(:697-701) under `torch.autocast("cuda", dtype=torch.bfloat16)`.

PINNED to the reference (tests/test_vae_oracle.py, fixtures tests/golden/vae_tiling.* produced by `gen_fixtures.py vae` running
the reference's own code in place): the tile-size rule (its constructor, autoencoder_kl_causal_3d.py:132-139), `blend_v` /
`blend_h` (:384-399) and the tile schedule + blended pixels of `spatial_tiled_decode` (:472-525, with a recording decoder
stand-in) -- bit for bit, and `mixgrpo_amd/vae.py`'s tiling is held to the same fixtures.
PARITY UNPINNED for everything inside a tile -- convolutions, GroupNorm, SiLU, the mid block's attention, upsampling, the
channel configuration: `AutoencoderKL` lives in diffusers (0.32.x per the reference's requirements), which is neither
importable here nor vendored, and the reference holds no fixture of a decoded image.  The block structure below follows the 2-D originals of the
blocks the reference DOES vendor in 3-D form for its Hunyuan VAE -- same lineage, same forward order:
  * ResnetBlock:  fastvideo/models/hunyuan/vae/unet_causal_3d_blocks.py:404-462  (norm1, silu, conv1, norm2, silu, conv2,
                  optional 1x1 conv_shortcut, (input + hidden) / output_scale_factor with factor 1)
  * mid block:    :667-693 (resnet, attention, resnet) with `Attention(heads = 1, dim_head = C, norm_num_groups, residual_connection,
                  bias)` built at :629-645
  * upsampler:    :148-206 (nearest 2x computed in fp32 for bf16 inputs, then a 3x3 conv)
  * up block:     :816-829 (resnets, then upsamplers)
  * decoder:      fastvideo/models/hunyuan/vae/vae.py:251-312 (conv_in, mid, up blocks, GroupNorm, SiLU, conv_out)
  * tiling:       fastvideo/models/hunyuan/vae/autoencoder_kl_causal_3d.py:132-139 (tile sizes), :384-399 (blend_v / blend_h,
                  IN PLACE on the later tile), :472-525 (spatial_tiled_decode), :338-342 (only when a side exceeds the tile)
and the FLUX.1-dev VAE configuration is recollected (16 latent channels, block_out_channels (128, 256, 512, 512), 2 layers per
block, 32 groups, sample_size 1024, scaling 0.3611, shift 0.1159 -- the two constants are in the reference, :286).

Rounding model of bf16 autocast with bf16 weights: a convolution / linear takes bf16 operands, accumulates in fp32 and returns
bf16; GroupNorm is an autocast-to-fp32 op whose fp32 result feeds SiLU in fp32; the next conv casts its input to bf16; residual
sums are bf16 + bf16 -> bf16; attention = softmax in fp32 on fp32 scores, probabilities cast to bf16 for the second product.
"""
from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn.functional as F


@dataclass
class VaeConfig:
    latent_channels: int = 16
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    sample_size: int = 1024
    scaling_factor: float = 0.3611
    shift_factor: float = 0.1159
    mid_block_add_attention: bool = True


def rb(t):
    return t.to(torch.bfloat16).float()


def param_shapes(cfg: VaeConfig):
    """diffusers key -> shape for the decoder half of AutoencoderKL."""
    ch = list(reversed(cfg.block_out_channels))
    out = {}

    def conv(name, co, ci, k):
        out[name + ".weight"] = (co, ci, k, k)
        out[name + ".bias"] = (co,)

    def norm(name, c):
        out[name + ".weight"] = (c,)
        out[name + ".bias"] = (c,)

    def resnet(name, ci, co):
        norm(name + ".norm1", ci)
        conv(name + ".conv1", co, ci, 3)
        norm(name + ".norm2", co)
        conv(name + ".conv2", co, co, 3)
        if ci != co:
            conv(name + ".conv_shortcut", co, ci, 1)

    conv("decoder.conv_in", ch[0], cfg.latent_channels, 3)
    resnet("decoder.mid_block.resnets.0", ch[0], ch[0])
    if cfg.mid_block_add_attention:
        a = "decoder.mid_block.attentions.0"
        norm(a + ".group_norm", ch[0])
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            out[f"{a}.{n}.weight"] = (ch[0], ch[0])
            out[f"{a}.{n}.bias"] = (ch[0],)
    resnet("decoder.mid_block.resnets.1", ch[0], ch[0])
    prev = ch[0]
    for i, co in enumerate(ch):
        for j in range(cfg.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else co, co)
        if i != len(ch) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co, 3)
        prev = co
    norm("decoder.conv_norm_out", ch[-1])
    conv("decoder.conv_out", cfg.out_channels, ch[-1], 3)
    return out


def init_params(cfg: VaeConfig, seed=0):
    """Random bf16-valued parameters (fp32 tensors holding bf16 values) at a scale that keeps activations O(1)."""
    g = torch.Generator().manual_seed(seed)
    P = {}
    for k, shp in param_shapes(cfg).items():
        if k.endswith(".weight") and len(shp) == 1:
            P[k] = rb(1.0 + 0.1 * torch.randn(shp, generator=g))
        elif k.endswith(".bias"):
            P[k] = rb(0.05 * torch.randn(shp, generator=g))
        else:
            fan_in = shp[1] * (shp[2] * shp[3] if len(shp) == 4 else 1)
            P[k] = rb(torch.randn(shp, generator=g) / fan_in ** 0.5)
    return P


def _conv(P, name, x, pad):
    return rb(F.conv2d(rb(x), P[name + ".weight"], P[name + ".bias"], padding=pad))


def _gn(P, name, x, G):
    return F.group_norm(x, G, P[name + ".weight"], P[name + ".bias"], eps=1e-6)          # fp32 out


def _resnet(P, name, x, G):
    h = _conv(P, name + ".conv1", F.silu(_gn(P, name + ".norm1", x, G)), 1)
    h = _conv(P, name + ".conv2", F.silu(_gn(P, name + ".norm2", h, G)), 1)
    sc = _conv(P, name + ".conv_shortcut", x, 0) if (name + ".conv_shortcut.weight") in P else x
    return rb(sc + h)


def _attention(P, name, x, G):
    B, C, H, W = x.shape
    n = _gn(P, name + ".group_norm", x, G).reshape(B, C, H * W).transpose(1, 2)           # [B, HW, C] fp32
    lin = lambda nm, t: rb(rb(t) @ P[f"{name}.{nm}.weight"].t() + P[f"{name}.{nm}.bias"])
    q, k, v = lin("to_q", n), lin("to_k", n), lin("to_v", n)
    s = (q @ k.transpose(1, 2)) * (C ** -0.5)                                              # one head of dim C
    p = rb(torch.softmax(s, dim=-1))
    o = lin("to_out.0", rb(p @ v))
    return rb(o.transpose(1, 2).reshape(B, C, H, W) + x)


def decoder(P, cfg: VaeConfig, z):
    """AutoencoderKL.decoder(z): z [B, latent_channels, h, w] fp32 -> image [B, 3, 8h, 8w] (bf16 values in fp32)."""
    G = cfg.norm_num_groups
    x = _conv(P, "decoder.conv_in", z, 1)
    x = _resnet(P, "decoder.mid_block.resnets.0", x, G)
    if cfg.mid_block_add_attention:
        x = _attention(P, "decoder.mid_block.attentions.0", x, G)
    x = _resnet(P, "decoder.mid_block.resnets.1", x, G)
    n = len(cfg.block_out_channels)
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            x = _resnet(P, f"decoder.up_blocks.{i}.resnets.{j}", x, G)
        if i != n - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = _conv(P, f"decoder.up_blocks.{i}.upsamplers.0.conv", x, 1)
    x = F.silu(_gn(P, "decoder.conv_norm_out", x, G))
    return _conv(P, "decoder.conv_out", x, 1)


def tile_sizes(cfg: VaeConfig):
    """(tile_sample_min_size, tile_latent_min_size, overlap factor) -- autoencoder_kl_causal_3d.py:132-139"""
    return cfg.sample_size, int(cfg.sample_size / (2 ** (len(cfg.block_out_channels) - 1))), 0.25


def blend_v(a, b, blend_extent):
    """IN PLACE on b, row by row in the tensors' dtype (autoencoder_kl_causal_3d.py:384-390)."""
    blend_extent = min(a.shape[-2], b.shape[-2], blend_extent)
    for y in range(blend_extent):
        b[..., y, :] = a[..., -blend_extent + y, :] * (1 - y / blend_extent) + b[..., y, :] * (y / blend_extent)
    return b


def blend_h(a, b, blend_extent):
    blend_extent = min(a.shape[-1], b.shape[-1], blend_extent)
    for x in range(blend_extent):
        b[..., x] = a[..., -blend_extent + x] * (1 - x / blend_extent) + b[..., x] * (x / blend_extent)
    return b


def tiled_decode(P, cfg: VaeConfig, z, decode_tile=None):
    """autoencoder_kl_causal_3d.py:472-525 in 2-D.  Tiles are bf16 tensors (the decoder's output dtype): the blend arithmetic
    rounds to bf16 after every product and sum, as torch does on bf16 tensors."""
    ts, tl, ov = tile_sizes(cfg)
    overlap_size = int(tl * (1 - ov))
    blend_extent = int(ts * ov)
    row_limit = ts - blend_extent
    dec = decode_tile or (lambda t: decoder(P, cfg, t))
    rows = []
    for i in range(0, z.shape[-2], overlap_size):
        row = []
        for j in range(0, z.shape[-1], overlap_size):
            row.append(dec(z[:, :, i:i + tl, j:j + tl]).to(torch.bfloat16))
        rows.append(row)
    result_rows = []
    for i, row in enumerate(rows):
        result_row = []
        for j, tile in enumerate(row):
            if i > 0:
                tile = blend_v(rows[i - 1][j], tile, blend_extent)
            if j > 0:
                tile = blend_h(row[j - 1], tile, blend_extent)
            result_row.append(tile[..., :row_limit, :row_limit])
        result_rows.append(torch.cat(result_row, dim=-1))
    return torch.cat(result_rows, dim=-2)


def decode(P, cfg: VaeConfig, z, use_tiling=True):
    """AutoencoderKL.decode: tiled only when a latent side exceeds the tile (autoencoder_kl_causal_3d.py:338-342)."""
    _, tl, _ = tile_sizes(cfg)
    if use_tiling and (z.shape[-1] > tl or z.shape[-2] > tl):
        return tiled_decode(P, cfg, z)
    return decoder(P, cfg, z).to(torch.bfloat16)


def decode_latents(P, cfg: VaeConfig, latents, h, w):
    """train_grpo_flux.py:284-288: unpack the [B, N, 64] rollout latents, un-scale, decode."""
    B = latents.shape[0]
    hh, ww = 2 * (int(h) // 16), 2 * (int(w) // 16)
    x = latents.view(B, hh // 2, ww // 2, latents.shape[-1] // 4, 2, 2).permute(0, 3, 1, 4, 2, 5).reshape(B, -1, hh, ww)
    return decode(P, cfg, (x.float() / cfg.scaling_factor) + cfg.shift_factor)
