"""FLUX VAE decode on the HIP kernels: the `vae` object of the reference's rollout,
    vae = AutoencoderKL.from_pretrained(path, subfolder="vae", torch_dtype=torch.bfloat16)       (train_grpo_flux.py:697-701)
    vae.enable_tiling(); image = vae.decode(latents, return_dict=False)[0]                        (:279-289)
-- diffusers' AutoencoderKL, decoder half only (the trainer never encodes).  Same constructor call, `.config`, `.enable_tiling()`,
`.decode(z, return_dict=False)`; weights from the published `vae/diffusion_pytorch_model.safetensors` (diffusers key names).

Layout: one image at a time, activations NHWC bf16; every 3x3 convolution is an implicit GEMM on the MFMA GEMM kernels over a
zero-bordered copy written by the GroupNorm + SiLU (or upsample) pass in front of it, residual sums ride the GEMM epilogue, the
mid block's single-head attention (H W <= 16384 tokens of dim 512) is two GEMMs around an fp32 row softmax.  All arithmetic is in
csrc/ (gemm.hip, vae.hip); this file sequences it.  Tiling follows the reference lineage's `spatial_tiled_decode`
(fastvideo/models/hunyuan/vae/autoencoder_kl_causal_3d.py:472-525): a 1024^2 image (128^2 latent) is exactly one tile and is
decoded whole; larger images are decoded tile by tile and blended (in place on the later tile, :384-399).
"""
import json
import os
from dataclasses import dataclass
from typing import Tuple

import torch

from . import ops
from .ops import BF16, F32, Rows, EPI_BIAS, EPI_BIAS_GATE_RES, EPI_F32_ACC


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


def _pad_ch(c):
    for p in (64, 128, 256, 512):
        if c <= p:
            return p
    raise ValueError(f"{c} channels: the convolution kernels take up to 512 per tap")


class AutoencoderKL:
    def __init__(self, config: VaeConfig = None, device="cuda"):
        self.config = config or VaeConfig()
        self.device = torch.device(device)
        self.use_tiling = False
        c = self.config
        self.tile_sample_min_size = c.sample_size
        self.tile_latent_min_size = int(c.sample_size / (2 ** (len(c.block_out_channels) - 1)))
        self.tile_overlap_factor = 0.25
        for ch in c.block_out_channels:
            if ch not in (64, 128, 256, 512):
                raise ValueError("block_out_channels must be 64, 128, 256 or 512 (channels per convolution tap)")
        self.P = None
        self._buf = {}
        self._ones = torch.ones(512, dtype=BF16, device=self.device)

    # ---------------------------------------------------------------------------------------------- weights
    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=None, device="cuda"):
        from safetensors.torch import load_file
        if subfolder and os.path.isdir(os.path.join(path, subfolder)):
            path = os.path.join(path, subfolder)
        with open(os.path.join(path, "config.json")) as f:
            raw = json.load(f)
        keys = set(VaeConfig.__dataclass_fields__)
        cfg = VaeConfig(**{k: (tuple(v) if k == "block_out_channels" else v) for k, v in raw.items()
                           if k in keys and v is not None})
        m = cls(cfg, device=device)
        m.load_state_dict(load_file(os.path.join(path, "diffusion_pytorch_model.safetensors")))
        return m

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def requires_grad_(self, flag=False):
        return self

    def load_state_dict(self, sd):
        """diffusers keys (`decoder.*`; encoder / quant-conv entries are ignored).  Conv weights OIHW -> [O][kh][kw][I] bf16 with
        the input channels padded to the kernel's tap width and conv_out's 3 output channels to 4."""
        c = self.config
        dev = self.device
        P = {}
        need = [k for k in _decoder_keys(c)]
        missing = [k for k in need if k not in sd]
        if missing:
            raise KeyError(f"VAE state dict lacks {len(missing)} decoder tensors, e.g. {missing[:3]}")
        for k in need:
            t = sd[k].to(dev)
            if t.dim() == 4 and t.shape[-1] == 3:
                O, I = t.shape[0], t.shape[1]
                Ip, Op = _pad_ch(I), (O + 3) // 4 * 4
                w = torch.zeros(Op, 3, 3, Ip, dtype=BF16, device=dev)
                w[:O, :, :, :I] = t.permute(0, 2, 3, 1).to(BF16)
                P[k] = w.contiguous()
            elif t.dim() == 4:                                   # 1x1 shortcut
                P[k] = t[:, :, 0, 0].to(BF16).contiguous()
            elif k.endswith("conv_out.bias"):
                b = torch.zeros((t.shape[0] + 3) // 4 * 4 + 4, dtype=BF16, device=dev)   # 16-byte sized
                b[:t.shape[0]] = t.to(BF16)
                P[k] = b
            else:
                P[k] = t.to(BF16).contiguous()
        a = "decoder.mid_block.attentions.0"
        if c.mid_block_add_attention:
            P[a + ".qkv.weight"] = torch.cat([P[f"{a}.to_{n}.weight"] for n in "qkv"], 0).contiguous()
            P[a + ".qkv.bias"] = torch.cat([P[f"{a}.to_{n}.bias"] for n in "qkv"], 0).contiguous()
        self.P = P
        return self

    def init_synthetic(self, seed=0):
        """Random bf16 weights of the configured architecture (no checkpoint offline); same generator as oracle/vae.py."""
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for k, shp in _decoder_shapes(self.config).items():
            if k.endswith(".weight") and len(shp) == 1:
                sd[k] = (1.0 + 0.1 * torch.randn(shp, generator=g)).to(BF16)
            elif k.endswith(".bias"):
                sd[k] = (0.05 * torch.randn(shp, generator=g)).to(BF16)
            else:
                fan_in = shp[1] * (shp[2] * shp[3] if len(shp) == 4 else 1)
                sd[k] = (torch.randn(shp, generator=g) / fan_in ** 0.5).to(BF16)
        return self.load_state_dict(sd)

    # ---------------------------------------------------------------------------------------------- buffers
    def _plain(self, tag, M, C, dtype=BF16):
        key = (tag, M, C, dtype)
        t = self._buf.get(key)
        if t is None:
            t = self._buf[key] = torch.empty(M, C, dtype=dtype, device=self.device)
        return t

    def _padded(self, H, W, C, tag="pad"):
        key = (tag, H, W, C)
        t = self._buf.get(key)
        if t is None:                                           # zero border = the convolutions' padding; interiors only are written
            t = self._buf[key] = torch.zeros(H + 2, W + 2, C, dtype=BF16, device=self.device)
        return t

    # ---------------------------------------------------------------------------------------------- blocks
    def _resnet(self, name, x, H, W, Cin, Cout):
        P, G = self.P, self.config.norm_num_groups
        pad = self._padded(H, W, Cin)
        ops.group_norm(x, P[name + ".norm1.weight"], P[name + ".norm1.bias"], pad, H, W, Cin, G, True, True)
        h = self._plain("h", H * W, Cout)
        ops.conv3x3(pad, P[name + ".conv1.weight"], P[name + ".conv1.bias"], h, H, W, Cin, Cout)
        pad2 = self._padded(H, W, Cout)
        ops.group_norm(h, P[name + ".norm2.weight"], P[name + ".norm2.bias"], pad2, H, W, Cout, G, True, True)
        if Cin != Cout:                                          # 1x1 conv_shortcut: a plain GEMM over the pixels
            y = self._plain("x", H * W, Cout)                    # another buffer than x: keyed by the channel count
            ops.gemm(Rows.of(x), P[name + ".conv_shortcut.weight"], P[name + ".conv_shortcut.bias"], Rows.of(y), Cout, Cin)
            x = y
        ops.conv3x3(pad2, P[name + ".conv2.weight"], P[name + ".conv2.bias"], x, H, W, Cout, Cout, ones=self._ones)
        return x

    def _attention(self, name, x, H, W, C):
        P, G = self.P, self.config.norm_num_groups
        M = H * W
        if M > 16384:
            raise ValueError("mid-block attention handles up to 16384 tokens (a 128 x 128 latent tile)")
        n = self._plain("attn_n", M, C)
        ops.group_norm(x, P[name + ".group_norm.weight"], P[name + ".group_norm.bias"], n, H, W, C, G, False, False)
        qkv = self._plain("attn_qkv", M, 3 * C)
        ops.gemm(Rows.of(n), P[name + ".qkv.weight"], P[name + ".qkv.bias"], Rows.of(qkv), 3 * C, C)
        S = self._plain("attn_s", M, M, F32)
        ops.gemm(Rows(qkv, M, 3 * C), qkv[0, C:], None, Rows.of(S), M, C, EPI_F32_ACC, beta=0.0, ldw=3 * C)     # q k^T, fp32
        Pm = self._plain("attn_p", M, M)
        ops.softmax_rows(S, Pm, M, M, float(C) ** -0.5)
        Mp = (M + 63) // 64 * 64
        vt = self._plain("attn_vt", C, Mp)
        if Mp != M:
            vt.zero_()
        ops.transpose(Rows(qkv[0, 2 * C:], M, 3 * C), C, vt, Mp)
        o = self._plain("attn_o", M, C)
        K = M if M % 64 == 0 else None
        if K is None:                                            # contraction length must be a multiple of 64: zero-padded P
            Pp = self._plain("attn_pp", M, Mp)
            Pp.zero_()
            Pp[:, :M].copy_(Pm)
            Pm, K = Pp, Mp
        ops.gemm(Rows.of(Pm), vt, None, Rows.of(o), C, K, ldw=Mp)
        ops.gemm(Rows.of(o), P[name + ".to_out.0.weight"], P[name + ".to_out.0.bias"], Rows.of(x), C, C, EPI_BIAS_GATE_RES,
                 gate=self._ones, gate_ld=0)
        return x

    def _decoder(self, z):
        """z [latent_channels, h, w] fp32 on the device -> image [out_channels, 8h, 8w] bf16"""
        c, P = self.config, self.P
        if P is None:
            raise RuntimeError("VAE weights not loaded")
        ch = list(reversed(c.block_out_channels))
        _, H, W = z.shape
        Cz = _pad_ch(c.latent_channels)
        # its own buffer: only the first `latent_channels` of its Cz channels are ever written, the rest must stay the zeros of
        # the allocation (conv_in's weight is zero there, but 0 * inf = NaN) -- the shared ("pad", H, W, 64) buffer of a
        # 64-channel layer at the same resolution would leave stale activations in them
        pad = self._padded(H, W, Cz, tag="pad_z")
        ops.latents_to_pad(z.contiguous(), pad, c.latent_channels, H, W, Cz)
        x = self._plain("x", H * W, ch[0])
        ops.conv3x3(pad, P["decoder.conv_in.weight"], P["decoder.conv_in.bias"], x, H, W, Cz, ch[0])
        x = self._resnet("decoder.mid_block.resnets.0", x, H, W, ch[0], ch[0])
        if c.mid_block_add_attention:
            x = self._attention("decoder.mid_block.attentions.0", x, H, W, ch[0])
        x = self._resnet("decoder.mid_block.resnets.1", x, H, W, ch[0], ch[0])
        prev = ch[0]
        for i, co in enumerate(ch):
            for j in range(c.layers_per_block + 1):
                x = self._resnet(f"decoder.up_blocks.{i}.resnets.{j}", x, H, W, prev if j == 0 else co, co)
            if i != len(ch) - 1:
                pad = self._padded(2 * H, 2 * W, co)
                ops.upsample2x_pad(x, pad, H, W, co)
                H, W = 2 * H, 2 * W
                y = self._plain("x", H * W, co)
                k = f"decoder.up_blocks.{i}.upsamplers.0.conv"
                ops.conv3x3(pad, P[k + ".weight"], P[k + ".bias"], y, H, W, co, co)
                x = y
            prev = co
        pad = self._padded(H, W, ch[-1])
        ops.group_norm(x, P["decoder.conv_norm_out.weight"], P["decoder.conv_norm_out.bias"], pad, H, W, ch[-1],
                       c.norm_num_groups, True, True)
        Co = (c.out_channels + 3) // 4 * 4
        y = self._plain("img_nhwc", H * W, 8)
        ops.conv3x3(pad, P["decoder.conv_out.weight"], P["decoder.conv_out.bias"], y, H, W, ch[-1], Co, ld_out=8)
        img = torch.empty(c.out_channels, H, W, dtype=BF16, device=self.device)
        ops.nhwc_to_image(y, 8, img, c.out_channels, H, W)
        return img

    # ---------------------------------------------------------------------------------------------- public surface
    def enable_tiling(self, use_tiling=True):
        self.use_tiling = use_tiling

    def disable_tiling(self):
        self.use_tiling = False

    def _decode_batch(self, z):
        return torch.stack([self._decoder(z[b].float()) for b in range(z.shape[0])])

    def tiled_decode(self, z):
        """autoencoder_kl_causal_3d.py:472-525 in 2-D; the blends are bf16 tensor arithmetic on the tiles, row by row, IN PLACE
        on the later tile exactly as :384-399 (the tile above / to the left has already been blended itself)."""
        tl, ts, ov = self.tile_latent_min_size, self.tile_sample_min_size, self.tile_overlap_factor
        overlap_size = int(tl * (1 - ov))
        blend_extent = int(ts * ov)
        row_limit = ts - blend_extent
        rows = []
        for i in range(0, z.shape[-2], overlap_size):
            rows.append([self._decode_batch(z[:, :, i:i + tl, j:j + tl]) for j in range(0, z.shape[-1], overlap_size)])
        result_rows = []
        for i, row in enumerate(rows):
            result_row = []
            for j, tile in enumerate(row):
                if i > 0:
                    tile = _blend(rows[i - 1][j], tile, blend_extent, -2)
                if j > 0:
                    tile = _blend(row[j - 1], tile, blend_extent, -1)
                result_row.append(tile[..., :row_limit, :row_limit])
            result_rows.append(torch.cat(result_row, dim=-1))
        return torch.cat(result_rows, dim=-2)

    @torch.no_grad()
    def decode(self, z, return_dict=True, generator=None):
        """z [B, latent_channels, h, w] (already un-scaled: `latents / scaling_factor + shift_factor`, :286) -> images
        [B, 3, 8h, 8w] bf16.  Tiled only when a latent side exceeds the tile (autoencoder_kl_causal_3d.py:338-342)."""
        z = z.to(self.device)
        tl = self.tile_latent_min_size
        if self.use_tiling and (z.shape[-1] > tl or z.shape[-2] > tl):
            img = self.tiled_decode(z)
        else:
            img = self._decode_batch(z)
        return {"sample": img} if return_dict else (img,)


def _blend(a, b, blend_extent, dim):
    """blend_v (dim -2) / blend_h (dim -1) of the reference lineage, vectorised over the blended rows.  Per element, as torch
    evaluates `a[y] * (1 - y / e) + b[y] * (y / e)` on bf16 tensors with python-float weights: each product is formed in fp32 with
    the fp32 weight and rounded to bf16, then the two bf16 products are added and rounded once more.  In place on b."""
    e = min(a.shape[dim], b.shape[dim], blend_extent)
    if e <= 0:
        return b
    shape = [1] * b.dim()
    shape[dim] = e
    wb = torch.tensor([y / e for y in range(e)], dtype=torch.float64)
    wa = (1.0 - wb).to(F32).to(b.device).view(shape)           # python computes 1 - y / e in double, torch takes it as fp32
    wb = wb.to(F32).to(b.device).view(shape)
    src = a.narrow(dim, a.shape[dim] - e, e)
    dst = b.narrow(dim, 0, e)
    dst.copy_((src.float() * wa).to(BF16) + (dst.float() * wb).to(BF16))
    return b


def _decoder_shapes(cfg):
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


def _decoder_keys(cfg):
    return list(_decoder_shapes(cfg))
