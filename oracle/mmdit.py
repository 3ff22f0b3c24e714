"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product.

PARITY UNPINNED: the FLUX MMDiT is `diffusers==0.32.2`'s FluxTransformer2DModel (pinned by the reference's
env_setup.sh:38; call sites fastvideo/utils/sampling_utils.py:68-82 and fastvideo/train_grpo_flux.py:134-144).
diffusers is neither in /root/reference nor installed here and the reference holds no test or fixture for it, so
this restatement follows the published architecture as recorded in SURVEY.md Appendix A, with the in-repo
structural witnesses listed there (fastvideo/models/hunyuan_hf/modeling_hunyuan.py:467-631,815-952,
fastvideo/models/hunyuan/modules/posemb_layers.py:140-193,267-314, embed_layers.py:99-123).

Plain PyTorch (fp32 storage), with the bf16 rounding points of `torch.autocast(bf16)` over fp32 master weights
made explicit (`_bf`): every Linear takes bf16-rounded inputs/weights/bias, accumulates in fp32 and rounds its
output to bf16; LayerNorm/RMSNorm/RoPE run in fp32; the residual stream is bf16.  Differentiable (the casts are
straight-through), so torch autograd of this file is the reference for the HIP backward.
State-dict keys use the diffusers names.
"""
import math
from dataclasses import dataclass, asdict
from typing import Dict, Tuple

import torch
import torch.nn.functional as F


@dataclass
class FluxConfig:
    patch_size: int = 1
    in_channels: int = 64
    num_layers: int = 19
    num_single_layers: int = 38
    attention_head_dim: int = 128
    num_attention_heads: int = 24
    joint_attention_dim: int = 4096
    pooled_projection_dim: int = 768
    guidance_embeds: bool = True
    axes_dims_rope: Tuple[int, int, int] = (16, 56, 56)

    @property
    def dim(self):
        return self.attention_head_dim * self.num_attention_heads

    def to_dict(self):
        d = asdict(self)
        d["axes_dims_rope"] = list(self.axes_dims_rope)
        return d


class _RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


def _bf(x):
    """Round to bf16 (value kept in fp32 storage); gradient passes straight through."""
    return _RoundBF16.apply(x.to(torch.float32))


def param_shapes(cfg: FluxConfig) -> Dict[str, Tuple[int, ...]]:
    d, hd = cfg.dim, cfg.attention_head_dim
    s = {}

    def lin(name, out_f, in_f):
        s[name + ".weight"] = (out_f, in_f)
        s[name + ".bias"] = (out_f,)

    lin("x_embedder", d, cfg.in_channels)
    lin("context_embedder", d, cfg.joint_attention_dim)
    embs = ["timestep_embedder", "text_embedder"] + (["guidance_embedder"] if cfg.guidance_embeds else [])
    for e in embs:
        lin(f"time_text_embed.{e}.linear_1", d, cfg.pooled_projection_dim if e == "text_embedder" else 256)
        lin(f"time_text_embed.{e}.linear_2", d, d)
    for i in range(cfg.num_layers):
        p = f"transformer_blocks.{i}"
        lin(f"{p}.norm1.linear", 6 * d, d)
        lin(f"{p}.norm1_context.linear", 6 * d, d)
        for n in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
            lin(f"{p}.attn.{n}", d, d)
        for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            s[f"{p}.attn.{n}.weight"] = (hd,)
        for ff in ("ff", "ff_context"):
            lin(f"{p}.{ff}.net.0.proj", 4 * d, d)
            lin(f"{p}.{ff}.net.2", d, 4 * d)
    for i in range(cfg.num_single_layers):
        p = f"single_transformer_blocks.{i}"
        lin(f"{p}.norm.linear", 3 * d, d)
        lin(f"{p}.proj_mlp", 4 * d, d)
        lin(f"{p}.proj_out", d, 5 * d)
        for n in ("to_q", "to_k", "to_v"):
            lin(f"{p}.attn.{n}", d, d)
        for n in ("norm_q", "norm_k"):
            s[f"{p}.attn.{n}.weight"] = (hd,)
    lin("norm_out.linear", 2 * d, d)
    lin("proj_out", cfg.patch_size * cfg.patch_size * cfg.in_channels, d)
    return s


def init_params(cfg: FluxConfig, seed: int = 0, std: float = 0.02, bias_std: float = 0.0, device="cpu"):
    """Synthetic weights: Linear weights N(0, std^2), biases N(0, bias_std^2) (0 -> zeros),
    RMSNorm weights 1 (+ small noise when bias_std>0 so tests see them)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = {}
    for k, shp in param_shapes(cfg).items():
        if k.endswith(".bias"):
            t = torch.randn(shp, generator=g) * bias_std if bias_std > 0 else torch.zeros(shp)
        elif len(shp) == 1:
            t = torch.ones(shp) + (torch.randn(shp, generator=g) * bias_std if bias_std > 0 else 0)
        else:
            t = torch.randn(shp, generator=g) * std
        out[k] = t.to(device)
    return out


def linear(x, w, b):
    """autocast Linear: bf16 operands, fp32 accumulate (+bias in fp32), one rounding to bf16."""
    y = _bf(x) @ _bf(w).t()
    if b is not None:
        y = y + _bf(b)
    return _bf(y)


def silu(x):
    return _bf(F.silu(x))


def gelu_tanh(x):
    return _bf(F.gelu(x, approximate="tanh"))


def layer_norm(x, eps=1e-6):
    return F.layer_norm(x, (x.shape[-1],), None, None, eps)


def modulate(x, shift, scale):
    """LN(x) [fp32] * (1+scale) [bf16, rounded] + shift [bf16] -> fp32 (next Linear rounds it)."""
    return layer_norm(x) * _bf(1 + scale)[:, None] + shift[:, None]


def rms_norm(x, w, eps=1e-6):
    var = x.pow(2).mean(-1, keepdim=True)
    return x * torch.rsqrt(var + eps) * w


def sincos256(t):
    """Timesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin], 128 frequencies."""
    half = 128
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    ang = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


def rope_tables(ids, axes_dims, theta=10000.0):
    """FluxPosEmbed: per axis a, freqs = theta^-(arange(0,dim,2)/dim) in float64, angle = pos (x) freqs,
    cos/sin repeat_interleave(2), concatenated over axes -> [S, head_dim] fp32."""
    cos, sin = [], []
    pos = ids.float()
    for a, dim in enumerate(axes_dims):
        fr = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float64, device=ids.device) / dim))
        ang = torch.outer(pos[:, a].double(), fr)
        cos.append(ang.cos().repeat_interleave(2, dim=1).float())
        sin.append(ang.sin().repeat_interleave(2, dim=1).float())
    return torch.cat(cos, -1), torch.cat(sin, -1)


def apply_rope(x, cos, sin):
    """x [B,H,S,D] fp32; interleaved pairs (x0,x1) -> (x0 cos - x1 sin, x1 cos + x0 sin)."""
    xr = x.reshape(*x.shape[:-1], -1, 2)
    rot = torch.stack([-xr[..., 1], xr[..., 0]], dim=-1).flatten(-2)
    return x * cos[None, None] + rot * sin[None, None]


def attention(q, k, v):
    """SDPA under autocast: bf16 q,k,v; scores/softmax in fp32; P rounded to bf16 before P@V; bf16 output."""
    q, k, v = _bf(q), _bf(k), _bf(v)
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(q.shape[-1]))
    p = torch.softmax(s, dim=-1)
    return _bf(p @ v)


def heads(x, H):
    B, S, D = x.shape
    return x.view(B, S, H, D // H).transpose(1, 2)


def unheads(x):
    B, H, S, hd = x.shape
    return x.transpose(1, 2).reshape(B, S, H * hd)


def model_time_inputs(timestep, guidance):
    """`timestep.to(hidden_states.dtype) * 1000` with hidden_states already bf16 (x_embedder ran under autocast):
    both the cast and the product are rounded to bf16 -- e.g. 0.954 -> 0.953125 -> 952."""
    t = _bf(_bf(timestep) * 1000)
    g = _bf(_bf(guidance) * 1000)
    return t, g


def forward(P, cfg: FluxConfig, hidden_states, encoder_hidden_states, timestep, guidance, txt_ids, pooled_projections,
            img_ids, collect=None):
    """FluxTransformer2DModel.forward restated.  Returns the velocity [B,N,in_channels] (bf16 values, fp32 storage).
    `collect` (dict) receives intermediates for per-stage tests."""
    H = cfg.num_attention_heads
    L = encoder_hidden_states.shape[1]
    B = hidden_states.shape[0]
    h = linear(hidden_states, P["x_embedder.weight"], P["x_embedder.bias"])
    t, g = model_time_inputs(timestep, guidance.expand(B) if guidance.numel() == 1 else guidance)

    def mlp_embed(name, x):
        y = linear(x, P[f"time_text_embed.{name}.linear_1.weight"], P[f"time_text_embed.{name}.linear_1.bias"])
        return linear(silu(y), P[f"time_text_embed.{name}.linear_2.weight"], P[f"time_text_embed.{name}.linear_2.bias"])

    temb = mlp_embed("timestep_embedder", _bf(sincos256(t)))
    if cfg.guidance_embeds:
        temb = _bf(temb + mlp_embed("guidance_embedder", _bf(sincos256(g))))
    temb = _bf(temb + mlp_embed("text_embedder", pooled_projections))
    c = linear(encoder_hidden_states, P["context_embedder.weight"], P["context_embedder.bias"])
    ids = torch.cat([txt_ids.float(), img_ids.float()], dim=0)
    cos, sin = rope_tables(ids, cfg.axes_dims_rope)
    st = silu(temb)
    if collect is not None:
        collect.update(temb=temb, x_embed=h, ctx_embed=c, cos=cos, sin=sin)

    def qkv(x, p, names, nq, nk):
        q = heads(linear(x, P[f"{p}.{names[0]}.weight"], P[f"{p}.{names[0]}.bias"]), H)
        k = heads(linear(x, P[f"{p}.{names[1]}.weight"], P[f"{p}.{names[1]}.bias"]), H)
        v = heads(linear(x, P[f"{p}.{names[2]}.weight"], P[f"{p}.{names[2]}.bias"]), H)
        return rms_norm(q, P[f"{p}.{nq}.weight"]), rms_norm(k, P[f"{p}.{nk}.weight"]), v

    def gated_residual(x, gate, y):
        return _bf(x + _bf(gate[:, None] * y))

    for i in range(cfg.num_layers):
        p = f"transformer_blocks.{i}"
        m = linear(st, P[f"{p}.norm1.linear.weight"], P[f"{p}.norm1.linear.bias"]).chunk(6, dim=1)
        mc = linear(st, P[f"{p}.norm1_context.linear.weight"], P[f"{p}.norm1_context.linear.bias"]).chunk(6, dim=1)
        q, k, v = qkv(modulate(h, m[0], m[1]), p + ".attn", ("to_q", "to_k", "to_v"), "norm_q", "norm_k")
        qc, kc, vc = qkv(modulate(c, mc[0], mc[1]), p + ".attn", ("add_q_proj", "add_k_proj", "add_v_proj"),
                         "norm_added_q", "norm_added_k")
        Q = apply_rope(torch.cat([qc, q], dim=2), cos, sin)
        K = apply_rope(torch.cat([kc, k], dim=2), cos, sin)
        o = unheads(attention(Q, K, torch.cat([vc, v], dim=2)))
        oc, oi = o[:, :L], o[:, L:]
        h = gated_residual(h, m[2], linear(oi, P[f"{p}.attn.to_out.0.weight"], P[f"{p}.attn.to_out.0.bias"]))
        c = gated_residual(c, mc[2], linear(oc, P[f"{p}.attn.to_add_out.weight"], P[f"{p}.attn.to_add_out.bias"]))

        def ff(x, name):
            y = gelu_tanh(linear(x, P[f"{p}.{name}.net.0.proj.weight"], P[f"{p}.{name}.net.0.proj.bias"]))
            return linear(y, P[f"{p}.{name}.net.2.weight"], P[f"{p}.{name}.net.2.bias"])

        h = gated_residual(h, m[5], ff(modulate(h, m[3], m[4]), "ff"))
        c = gated_residual(c, mc[5], ff(modulate(c, mc[3], mc[4]), "ff_context"))
        if collect is not None:
            collect[f"double{i}_h"] = h
            collect[f"double{i}_c"] = c

    x = torch.cat([c, h], dim=1)
    for i in range(cfg.num_single_layers):
        p = f"single_transformer_blocks.{i}"
        m = linear(st, P[f"{p}.norm.linear.weight"], P[f"{p}.norm.linear.bias"]).chunk(3, dim=1)
        n = modulate(x, m[0], m[1])
        mlp = gelu_tanh(linear(n, P[f"{p}.proj_mlp.weight"], P[f"{p}.proj_mlp.bias"]))
        q, k, v = qkv(n, p + ".attn", ("to_q", "to_k", "to_v"), "norm_q", "norm_k")
        o = unheads(attention(apply_rope(q, cos, sin), apply_rope(k, cos, sin), v))
        y = linear(torch.cat([o, mlp], dim=2), P[f"{p}.proj_out.weight"], P[f"{p}.proj_out.bias"])
        x = gated_residual(x, m[2], y)
        if collect is not None:
            collect[f"single{i}_x"] = x

    h = x[:, L:]
    e = linear(st, P["norm_out.linear.weight"], P["norm_out.linear.bias"])
    scale, shift = e.chunk(2, dim=1)
    return linear(modulate(h, shift, scale), P["proj_out.weight"], P["proj_out.bias"])


class OracleFlux(torch.nn.Module):
    """nn.Module wrapper with the FLUX call signature (kwargs of sampling_utils.py:68-82); returns (bf16,)."""

    def __init__(self, cfg: FluxConfig, params: Dict[str, torch.Tensor]):
        super().__init__()
        self.cfg = cfg
        self.config = cfg.to_dict()
        self._names = list(params)
        self.params = torch.nn.ParameterList([torch.nn.Parameter(params[k].clone().float()) for k in self._names])

    def P(self):
        return dict(zip(self._names, self.params))

    def forward(self, hidden_states, encoder_hidden_states, timestep, guidance, txt_ids, pooled_projections, img_ids,
                joint_attention_kwargs=None, return_dict=False):
        out = forward(self.P(), self.cfg, hidden_states.float(), encoder_hidden_states.float(), timestep.float(),
                      guidance.float(), txt_ids, pooled_projections.float(), img_ids)
        return (out.to(torch.bfloat16),)

    def clip_grad_norm_(self, max_norm):
        return torch.nn.utils.clip_grad_norm_(self.parameters(), max_norm)
