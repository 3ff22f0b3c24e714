"""FLUX MMDiT on MI355X: drop-in for `diffusers.FluxTransformer2DModel` as the reference calls it
(fastvideo/utils/sampling_utils.py:68-82, fastvideo/train_grpo_flux.py:134-144,606,677-679).

Same call signature (kwargs), `.train()/.eval()/.parameters()/.state_dict()/.load_state_dict()/.config/
.clip_grad_norm_()`, diffusers state-dict key names.  All arithmetic runs in the hand-written HIP kernels of
csrc/ through the C ABI; this file is host orchestration: the flat parameter store, workspace reuse, the block
schedule and (for training) block-level activation recompute in the backward pass.

Precision policy = the reference's autocast(bf16) over fp32 master weights: bf16 compute copy of the weights,
fp32 MFMA accumulation, bf16 residual stream, fp32 norm statistics, fp32 gradients for weights.

Memory layout: ONE joint residual buffer X[B, S, d] (text rows first, S = L + N); the text / image streams of
the double blocks are row-batched views of it (no concat/split copies).  Parameters live in one flat fp32
buffer (+ a bf16 mirror with the identical element order) in which to_q|to_k|to_v are adjacent, so the fused
QKV projection is a view.
"""
import json
import math
import os
from dataclasses import asdict, dataclass
from typing import Dict, List, Tuple

import torch

from . import ops
from ._lib import MgxError
from .ops import BF16, F32, Rows, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_GATE_RES


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
        d["_class_name"] = "FluxTransformer2DModel"
        return d


def param_layout(cfg: FluxConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """Flat order of the parameters (diffusers names).  q|k|v weights (and biases) are adjacent on purpose."""
    d, hd = cfg.dim, cfg.attention_head_dim
    out = []

    def lin(name, o, i):
        out.append((name + ".weight", (o, i)))
        out.append((name + ".bias", (o,)))

    def qkv(prefix, names):
        for n in names:
            out.append((f"{prefix}.{n}.weight", (d, d)))
        for n in names:
            out.append((f"{prefix}.{n}.bias", (d,)))

    lin("x_embedder", d, cfg.in_channels)
    lin("context_embedder", d, cfg.joint_attention_dim)
    embs = ["timestep_embedder"] + (["guidance_embedder"] if cfg.guidance_embeds else []) + ["text_embedder"]
    for e in embs:
        lin(f"time_text_embed.{e}.linear_1", d, cfg.pooled_projection_dim if e == "text_embedder" else 256)
        lin(f"time_text_embed.{e}.linear_2", d, d)
    for i in range(cfg.num_layers):
        p = f"transformer_blocks.{i}"
        lin(f"{p}.norm1.linear", 6 * d, d)
        lin(f"{p}.norm1_context.linear", 6 * d, d)
        qkv(f"{p}.attn", ("to_q", "to_k", "to_v"))
        qkv(f"{p}.attn", ("add_q_proj", "add_k_proj", "add_v_proj"))
        lin(f"{p}.attn.to_out.0", d, d)
        lin(f"{p}.attn.to_add_out", d, d)
        for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            out.append((f"{p}.attn.{n}.weight", (hd,)))
        for ff in ("ff", "ff_context"):
            lin(f"{p}.{ff}.net.0.proj", 4 * d, d)
            lin(f"{p}.{ff}.net.2", d, 4 * d)
    for i in range(cfg.num_single_layers):
        p = f"single_transformer_blocks.{i}"
        lin(f"{p}.norm.linear", 3 * d, d)
        # to_q | to_k | to_v | proj_mlp adjacent (weights, then biases): one [7d, d] operand for the backward GEMMs
        for n in ("attn.to_q", "attn.to_k", "attn.to_v"):
            out.append((f"{p}.{n}.weight", (d, d)))
        out.append((f"{p}.proj_mlp.weight", (4 * d, d)))
        for n in ("attn.to_q", "attn.to_k", "attn.to_v"):
            out.append((f"{p}.{n}.bias", (d,)))
        out.append((f"{p}.proj_mlp.bias", (4 * d,)))
        lin(f"{p}.proj_out", d, 5 * d)
        for n in ("norm_q", "norm_k"):
            out.append((f"{p}.attn.{n}.weight", (hd,)))
    lin("norm_out.linear", 2 * d, d)
    lin("proj_out", cfg.patch_size * cfg.patch_size * cfg.in_channels, d)
    return out


class ParamStore:
    """Flat fp32 master + bf16 mirror (+ fp32 grad / Adam moments on demand), addressed by diffusers names."""

    ALIGN = 64

    def __init__(self, cfg: FluxConfig, device):
        self.cfg = cfg
        self.device = torch.device(device)
        self.index: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        off = 0
        for name, shape in param_layout(cfg):
            self.index[name] = (off, shape)
            n = 1
            for s in shape:
                n *= s
            off += (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = off
        self.w32 = torch.zeros(off, dtype=F32, device=self.device)
        self.w16 = torch.zeros(off, dtype=BF16, device=self.device)
        self.g32 = None

    def view(self, buf, name):
        off, shape = self.index[name]
        n = 1
        for s in shape:
            n *= s
        return buf[off:off + n].view(shape)

    def fused(self, buf, first, rows_total):
        """View of `rows_total` rows starting at tensor `first` (adjacent tensors of equal width)."""
        off, shape = self.index[first]
        if len(shape) == 1:
            return buf[off:off + rows_total]
        return buf[off:off + rows_total * shape[1]].view(rows_total, shape[1])

    def block_ranges(self):
        """{block prefix: (lo, hi)} element ranges of the flat buffers, in layout order (transformer blocks; everything
        before / after them under "head" / "tail").  Contiguous and 64-aligned: the buckets of the DP gradient reduction."""
        names = list(self.index)
        key = lambda n: ".".join(n.split(".")[:2]) if n.startswith(("transformer_blocks.", "single_transformer_blocks.")) \
            else ("head" if self.index[n][0] < self.index["norm_out.linear.weight"][0] else "tail")
        out, cur, lo = {}, None, 0
        for n in names:
            k = key(n)
            if k != cur:
                if cur is not None:
                    out[cur] = (lo, self.index[n][0])
                cur, lo = k, self.index[n][0]
        out[cur] = (lo, self.numel)
        return out

    def sync_bf16(self):
        self.w16.copy_(self.w32)

    def ensure_grad(self):
        if self.g32 is None:
            self.g32 = torch.zeros(self.numel, dtype=F32, device=self.device)
        return self.g32

    def init_synthetic(self, seed=0, std=0.02, bias_std=0.0):
        """Random-init weights of the right architecture (no checkpoints offline): N(0, std^2) matrices, zero (or
        N(0, bias_std^2)) biases, RMSNorm weights 1.  Generated on the device, tensor by tensor."""
        g = torch.Generator(device=self.device).manual_seed(seed)
        for name, (off, shape) in self.index.items():
            v = self.view(self.w32, name)
            if name.endswith(".bias"):
                if bias_std > 0:
                    v.normal_(0.0, bias_std, generator=g)
                else:
                    v.zero_()
            elif len(shape) == 1:
                v.fill_(1.0)
                if bias_std > 0:
                    v.add_(torch.randn(shape, generator=g, device=self.device) * bias_std)
            else:
                v.normal_(0.0, std, generator=g)
        self.sync_bf16()


def rope_tables(ids, axes_dims, theta=10000.0):
    """FluxPosEmbed tables [S, head_dim] fp32 (angles in fp64), computed once per id set on the device."""
    cos, sin = [], []
    pos = ids.float()
    for a, dim in enumerate(axes_dims):
        fr = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float64, device=ids.device) / dim))
        ang = torch.outer(pos[:, a].double(), fr)
        cos.append(ang.cos().repeat_interleave(2, dim=1).float())
        sin.append(ang.sin().repeat_interleave(2, dim=1).float())
    return torch.cat(cos, -1).contiguous(), torch.cat(sin, -1).contiguous()


class _Work:
    """Activation workspace for one (L, N) problem shape at a batch CAPACITY B; calls with a smaller batch use leading
    slices of it (`view`), so the rollout (B = G), the shared first step (B = 1) and the replay micro-batches share one
    allocation instead of one per batch size."""

    def __init__(self, cfg, B, L, N, device):
        d, H, hd = cfg.dim, cfg.num_attention_heads, cfg.attention_head_dim
        S = L + N
        self.B, self.L, self.N, self.S = B, L, N, S
        self.Sp = (S + 63) // 64 * 64
        e = lambda *shape, dtype=BF16: torch.empty(*shape, dtype=dtype, device=device)
        self.X = e(B, S, d)
        self.nrm = e(B * S, d)
        self.qkv = e(B * S, 3 * d)
        self.Q = e(B, H, S, hd)
        self.K = e(B, H, S, hd)
        self.Vt = torch.zeros(B, H, hd, self.Sp, dtype=BF16, device=device)   # padding must stay finite
        self.O = e(B, S, d)
        self.hid = e(B * S, 4 * d)
        self.cat = e(B, S, 5 * d)
        self.in16 = e(B * N, cfg.in_channels)
        self.out = e(B, N, cfg.patch_size * cfg.patch_size * cfg.in_channels)
        self.lse = e(B, H, S, dtype=F32)
        self.train = None                         # flux_backward._Train, at its own batch capacity
        self._f8 = None                           # (Q8, K8, V8t, amax): allocated when attention_dtype == "fp8"
        self._dims = (H, hd)

    def fp8_operands(self):
        """e4m3 copies of Q, K and the key-permuted V^T plus the per-(batch, head) amax table, at batch capacity."""
        if self._f8 is None:
            H, hd = self._dims
            u8 = lambda *shape: torch.empty(*shape, dtype=torch.uint8, device=self.X.device)
            self._f8 = (u8(self.B, H, self.S, hd), u8(self.B, H, self.S, hd), u8(self.B, H, hd, self.Sp),
                        torch.empty(3 * self.B * H, dtype=F32, device=self.X.device))
        return self._f8

    def view(self, B):
        return self if B == self.B else _WorkView(self, B)


class _WorkView:
    """The first B batches of a `_Work` (all buffers are batch-major and contiguous, so these are plain prefixes)."""

    def __init__(self, base, B):
        assert B <= base.B
        self.base, self.B, self.L, self.N, self.S, self.Sp = base, B, base.L, base.N, base.S, base.Sp
        M = B * base.S
        self.X, self.Q, self.K, self.Vt, self.O, self.cat = (t[:B] for t in (base.X, base.Q, base.K, base.Vt, base.O, base.cat))
        self.nrm, self.qkv, self.hid = base.nrm[:M], base.qkv[:M], base.hid[:M]
        self.in16, self.out, self.lse = base.in16[:B * base.N], base.out[:B], base.lse[:B]

    def fp8_operands(self):
        q8, k8, v8t, amax = self.base.fp8_operands()
        return q8[:self.B], k8[:self.B], v8t[:self.B], amax      # amax is laid out [3][B*H] for the CURRENT batch

    @property
    def train(self):
        return self.base.train

    @train.setter
    def train(self, v):
        self.base.train = v


class FluxTransformer2DModel(torch.nn.Module):
    """MI355X-native FLUX MMDiT.  `hipflux = FluxTransformer2DModel(FluxConfig(), device="cuda")`."""

    def __init__(self, config: FluxConfig = None, device="cuda", attention_dtype="bf16", **cfg_kwargs):
        super().__init__()
        if attention_dtype not in ("bf16", "fp8"):
            raise ValueError(f"attention_dtype {attention_dtype!r} is not supported (bf16 | fp8)")
        self.attention_dtype = attention_dtype    # "fp8": e4m3 MFMA attention forward (BASELINE.json configs[4])
        self.cfg = config or FluxConfig(**cfg_kwargs)
        self.config = self.cfg.to_dict()
        dev = torch.device(device)
        if dev.type != "cuda":
            raise MgxError("FluxTransformer2DModel (mixgrpo_amd) runs on an MI355X only; there is no CPU path")
        self.store = ParamStore(self.cfg, dev)
        # one flat fp32 parameter (what optimizers / clip_grad_norm_ see); named views via state_dict()
        self.flat_param = torch.nn.Parameter(self.store.w32, requires_grad=True)
        self._work: Dict[Tuple[int, int], _Work] = {}
        self._rope_cache = {}
        self.recompute = True

    # ------------------------------------------------------------------ parameters / checkpoints
    def state_dict(self, *args, **kwargs):
        return {k: self.store.view(self.store.w32, k) for k in self.store.index}

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self.store.index if k not in sd]
        unexpected = [k for k in sd if k not in self.store.index]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:5]} unexpected {unexpected[:5]}")
        with torch.no_grad():
            for k, v in sd.items():
                if k in self.store.index:
                    self.store.view(self.store.w32, k).copy_(v.to(F32))
        self.store.sync_bf16()
        return missing, unexpected

    def init_synthetic(self, seed=0, std=0.02, bias_std=0.0):
        self.store.init_synthetic(seed, std, bias_std)
        return self

    def W(self, name):
        return self.store.view(self.store.w16, name)

    def W32(self, name):
        return self.store.view(self.store.w32, name)

    def clip_grad_norm_(self, max_norm):
        """Global L2 norm of the fp32 gradients, scaled in place like torch's clip_grad_norm_ (reference :606)."""
        g = self.store.ensure_grad()
        nsq = torch.zeros(1, dtype=F32, device=g.device)
        ops.sqnorm(g, nsq)
        total = nsq.sqrt()
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        g.mul_(coef)
        return total.squeeze(0)

    def grow_ff_keep(self, reserve_gib=None):
        """Spend the device memory that is still free (beyond a reserve) on kept FF pre-activations and QKV outputs, so that
        the training pass's recompute skips those blocks' d -> 4d / d -> 3d GEMMs (flux_backward.KEEP_FF / KEEP_QKV =
        "auto").  Called by `train_one_step` at the start of its second step, when the first has shown the step's real peak.
        Returns the number of blocks whose FF pre-activation is kept."""
        from . import flux_backward as FB
        if not FB.KEEP_ACTS or (FB.KEEP_FF != "auto" and FB.KEEP_QKV != "auto"):
            return self.ff_blocks_kept()
        reserve = (FB.KEEP_FF_RESERVE_GIB if reserve_gib is None else reserve_gib) * 2.0 ** 30
        for base in self._work.values():
            tr = base.train
            if tr is None or tr.keep is None:
                continue
            free, _total = torch.cuda.mem_get_info(self.store.device)
            tr.grow_auto(int(free - reserve))
        return self.ff_blocks_kept()

    def ff_blocks_kept(self):
        return sum(b.train.ff_kept() for b in self._work.values() if b.train is not None and b.train.keep is not None)

    def qkv_blocks_kept(self):
        return sum(b.train.qkv_kept() for b in self._work.values() if b.train is not None and b.train.keep is not None)

    # ------------------------------------------------------------------ forward
    def _workspace(self, B, L, N):
        """Workspace view for batch B of the (L, N) shape; the base grows to the largest batch seen."""
        key = (L, N)
        base = self._work.get(key)
        if base is None or base.B < B:
            if base is not None:
                self._work.pop(key)
                del base
            elif len(self._work) >= 2:   # keep at most two sequence shapes resident
                self._work.pop(next(iter(self._work)))
            torch.cuda.empty_cache()     # hand the outgrown buffers back before allocating the larger ones
            base = _Work(self.cfg, B, L, N, self.store.device)
            self._work[key] = base
        return base.view(B)

    def _attn(self, w, O, lse, ldo, o_bstride):
        """Joint attention of the current Q / K / Vt workspace into O (+ LSE).  `attention_dtype == "fp8"` (BASELINE.json
        configs[4]) quantises the operands per (batch, head) to e4m3 and runs both contractions on fp8 MFMA; every
        forward of the model (rollout, training forward) then uses it, the backward stays bf16."""
        H = self.cfg.num_attention_heads
        scale = self.attn_scale()
        if self.attention_dtype == "fp8":
            q8, k8, v8t, amax = w.fp8_operands()
            ops.attn_fp8_quantize(w.Q, w.K, w.Vt, q8, k8, v8t, amax, w.B, H, w.S, w.Sp)
            ops.attn_fwd_fp8(q8, k8, v8t, amax, O, lse, w.B, H, w.S, w.Sp, ldo, o_bstride, scale)
        elif ops.Q_PRESCALE:
            ops.attn_fwd_log2(w.Q, w.K, w.Vt, O, lse, w.B, H, w.S, w.Sp, ldo, o_bstride)
        else:
            ops.attn_fwd(w.Q, w.K, w.Vt, O, lse, w.B, H, w.S, w.Sp, ldo, o_bstride, scale)

    def _qkv_nograd(self, nrm, first, qkv, w, rows, s0, wq, wk, cos, sin):
        """Fused q | k | v projection of one stream ([tokens, d] -> `qkv` [tokens, 3d]) + QK-norm / RoPE / head split for a
        forward that keeps nothing (the rollout).  When the persistent GEMM takes the shapes: the value projection is issued
        with the operand roles swapped and lands transposed in `w.Vt` [B, H, 128, Sp] at sequence offset `s0`
        (`mgx_linear_bf16_t`), and the q | k projection normalises, rotates and head-splits its tile in the epilogue
        (`mgx_linear_qk_norm_rope`: `qkv` is not touched at all).  Otherwise the plain projection(s) and the norm pass."""
        d, H = self.cfg.dim, self.cfg.num_attention_heads
        Wf = self.store.fused(self.store.w16, f"{first}.weight", 3 * d)
        bf = self.store.fused(self.store.w16, f"{first}.bias", 3 * d)
        tokens = nrm.numel() // d
        B = tokens // rows
        vt = ops.LINEAR_VT and ops.linear_t(nrm, Wf[2 * d:], bf[2 * d:], w.Vt.view(-1)[s0:], tokens, d, d, w.Sp, rows, d * w.Sp)
        if vt and ops.LINEAR_QKNORM and H * 128 == d and ops.linear_qk_norm_rope(
                nrm, Wf[:2 * d], bf[:2 * d], wq, wk, cos, sin, w.Q, w.K, B, H, w.S, rows, s0, d, q_scale=self.q_scale(),
                pairs=self._rope_pairs(cos, sin)):
            return
        if vt:
            ops.gemm(Rows.of(nrm), Wf[:2 * d], bf[:2 * d], Rows(qkv, tokens, 3 * d), 2 * d, d)
        else:
            ops.gemm(Rows.of(nrm), Wf, bf, Rows.of(qkv), 3 * d, d)
        ops.qk_norm_rope(qkv, wq, wk, cos, sin, w.Q, w.K, None if vt else w.Vt, B, H, w.S, w.Sp, rows, s0, q_scale=self.q_scale())

    def q_scale(self):
        """What `mgx_qk_norm_rope_fwd_qs` multiplies q by before its one bf16 rounding: softmax scale * log2(e) (the scores
        are exponents of two: mgx_attn_fwd_log2), or 1 under MGX_ATTN_Q_PRESCALE=0."""
        return ops.LOG2E / math.sqrt(self.cfg.attention_head_dim) if ops.Q_PRESCALE else 1.0

    def attn_scale(self):
        """The `scale` every other consumer of that Q takes (mgx_attn_bwd, mgx_attn_fwd_fp8): ln 2 for the prescaled Q."""
        return ops.LN2 if ops.Q_PRESCALE else 1.0 / math.sqrt(self.cfg.attention_head_dim)

    def _rope(self, txt_ids, img_ids):
        key = (id(txt_ids), txt_ids._version, tuple(txt_ids.shape), id(img_ids), img_ids._version, tuple(img_ids.shape))
        hit = self._rope_cache.get("k")
        if hit is not None and hit[0] == key:
            return hit[1], hit[2]
        ids = torch.cat([txt_ids.float(), img_ids.float()], dim=0)
        cos, sin = rope_tables(ids, self.cfg.axes_dims_rope)
        self._rope_cache["k"] = (key, cos, sin, txt_ids, img_ids)   # keep the id tensors alive with the key
        return cos, sin

    def _rope_pairs(self, cos, sin):
        """[S, 64, 2] (cos, sin) per rotation pair for `mgx_linear_qk_norm_rope` when the tables repeat every pair's entry
        (FluxPosEmbed's do), else None; cached with the tables (MGX_ROPE_PAIR_TABLE=0: never)."""
        if not ops.ROPE_PAIR_TABLE:
            return None
        hit = self._rope_cache.get("pairs")
        if hit is None or hit[0] is not cos:
            hit = (cos, ops.rope_pair_table(cos, sin))
            self._rope_cache["pairs"] = hit
        return hit[1]

    def _temb(self, B, timestep, guidance, pooled, keep=None):
        """temb = t_emb + g_emb + text_emb (bf16 after every op), st = silu(temb)."""
        d = self.cfg.dim
        dev = self.store.device
        t = (timestep.to(BF16) * 1000).float().contiguous()          # `.to(hidden.dtype) * 1000` under autocast
        gd = guidance.to(dev)
        gd = (gd.to(BF16) * 1000).float()
        gd = gd.expand(B).contiguous() if gd.numel() == 1 else gd.contiguous()

        def mlp(name, x, K):
            h1 = torch.empty(B, d, dtype=BF16, device=dev)
            ops.skinny_linear(x, self.W(f"time_text_embed.{name}.linear_1.weight"),
                              self.W(f"time_text_embed.{name}.linear_1.bias"), h1, d, K)
            a1 = torch.empty_like(h1)
            ops.ew(h1, None, a1, 0)
            h2 = torch.empty_like(h1)
            ops.skinny_linear(a1, self.W(f"time_text_embed.{name}.linear_2.weight"),
                              self.W(f"time_text_embed.{name}.linear_2.bias"), h2, d, d)
            if keep is not None:
                keep[name] = (x, h1, a1)
            return h2

        te = torch.empty(B, 256, dtype=BF16, device=dev)
        ops.sincos_embed(t, te)
        temb = mlp("timestep_embedder", te, 256)
        if self.cfg.guidance_embeds:
            ge = torch.empty(B, 256, dtype=BF16, device=dev)
            ops.sincos_embed(gd, ge)
            gemb = mlp("guidance_embedder", ge, 256)
            t2 = torch.empty_like(temb)
            ops.ew(temb, gemb, t2, 2)
            temb = t2
        pe = mlp("text_embedder", pooled.to(BF16).contiguous(), self.cfg.pooled_projection_dim)
        t3 = torch.empty_like(temb)
        ops.ew(temb, pe, t3, 2)
        st = torch.empty_like(t3)
        ops.ew(t3, None, st, 0)
        return t3, st

    def _stream_rows(self, t, w, which, width):
        """Row-batched view of the text ('txt') or image ('img') rows of a joint [B, S, width] buffer."""
        if which == "txt":
            return Rows(t, w.B * w.L, width, w.L, w.S * width)
        return Rows(t[0, w.L:], w.B * w.N, width, w.N, w.S * width)

    def _double_block(self, i, w, st, cos, sin, save=None, mods_in=None, keep=None, replay=False, x_in=None):
        """`save`: keep the block's intermediates for the backward walk (recompute pass).  `keep`: per-block buffers
        {O, lse, y_attn, y_ff, x_mid} written by the training forward; with `replay` the recompute pass reads them
        back instead of re-running attention and the two output projections (to_out / ff.net.2), and reads the block's
        input from `x_in` (the saved block input: the residual stream is not written in this mode, so no copy of it)."""
        cfg, d, H = self.cfg, self.cfg.dim, self.cfg.num_attention_heads
        p = f"transformer_blocks.{i}"
        B = w.B
        dev = self.store.device
        mods = {}
        # text stream first: its rows come first in the stacked scratch buffers (nrm / qkv / hid / kept activations) and in
        # the joint [B, S, d] residual buffer, and its weights first in the parameter store -- problem 1 of the pair launches
        streams = (("txt", "norm1_context", ("add_q_proj", "add_k_proj", "add_v_proj"), "norm_added_q", "norm_added_k",
                    "to_add_out", "ff_context", w.L, 0),
                   ("img", "norm1", ("to_q", "to_k", "to_v"), "norm_q", "norm_k", "to_out.0", "ff", w.N, w.L))
        row0 = {"txt": 0, "img": w.B * w.L}     # row offsets inside the per-stream scratch buffers
        sl = {n: slice(row0[n], row0[n] + B * r) for n, r in (("txt", w.L), ("img", w.N))}
        W16, fused = self.W, self.store.fused
        qkv_kept = keep is not None and "qkv" in keep           # QKV output kept by the forward: no GEMM in the recompute
        qkv_buf = keep["qkv"] if qkv_kept else w.qkv
        nrm1 = w.nrm if save is None else save["nrm1"]
        for name, norm, qkvn, nq, nk, outn, ffn, rows, s0 in streams:
            if mods_in is not None:
                m = mods_in[name]
            else:
                m = torch.empty(B, 6 * d, dtype=BF16, device=dev)
                ops.skinny_linear(st, W16(f"{p}.{norm}.linear.weight"), W16(f"{p}.{norm}.linear.bias"), m, 6 * d, d)
            mods[name] = m
            Xs = self._stream_rows(w.X if x_in is None else x_in, w, name, d)
            ops.ln_modulate(Xs, m[:, 0:d], m[:, d:2 * d], 6 * d, nrm1[sl[name]], d)
        nograd = save is None and keep is None
        if nograd:
            # no-grad forward (the rollout): V^T straight from the value projection, QK-norm / RoPE in the q | k projection's epilogue
            for name, _, qkvn, nq, nk, _, _, rows, s0 in streams:
                self._qkv_nograd(nrm1[sl[name]], f"{p}.attn.{qkvn[0]}", qkv_buf[sl[name]], w, rows, s0,
                                 self.W32(f"{p}.attn.{nq}.weight"), self.W32(f"{p}.attn.{nk}.weight"), cos, sin)
        elif not (replay and qkv_kept):
            # both streams' fused QKV projections in one launch (text rows ride the image stream's rounds)
            ops.gemm_pair(*(x for name, _, qkvn, *_ in streams for x in (
                Rows.of(nrm1[sl[name]]), fused(self.store.w16, f"{p}.attn.{qkvn[0]}.weight", 3 * d),
                fused(self.store.w16, f"{p}.attn.{qkvn[0]}.bias", 3 * d), Rows.of(qkv_buf[sl[name]]))), 3 * d, d)
        for name, norm, qkvn, nq, nk, outn, ffn, rows, s0 in (() if nograd else streams):
            ops.qk_norm_rope(qkv_buf[sl[name]], self.W32(f"{p}.attn.{nq}.weight"), self.W32(f"{p}.attn.{nk}.weight"), cos, sin,
                             w.Q, w.K, w.Vt, B, H, w.S, w.Sp, rows, s0, q_scale=self.q_scale(),
                             **({} if save is None else dict(V=save["V"], Qt=save["Qt"], Kt=save["Kt"])))
        # the attention output lives in the block's keep buffer when there is one (written here, read by the backward):
        # no copy between the workspace and the kept tensor
        O_buf = keep["O"] if keep is not None else w.O
        if not replay:
            self._attn(w, O_buf, keep["lse"] if keep is not None else (w.lse if save is not None else None), d, w.S * d)
        aux1 = aux2 = hpre = None
        if keep is not None and not replay:
            aux1, aux2 = keep["y_attn"], keep["y_ff"]
        elif save is not None:
            aux1, aux2 = save["y_attn"], save["y_ff"]
        ff_kept = keep is not None and "hid_pre" in keep
        if save is not None:
            hpre = save["hid_pre"]                              # (with ff_kept this IS the kept buffer)
        elif ff_kept:
            hpre = keep["hid_pre"]
        part = lambda t, name: None if t is None else t[sl[name]]
        if not replay:
            # attention output projections, both streams: x += gate_msa * (O @ W^T + b)
            ops.gemm_pair(*(x for name, _, _, _, _, outn, *_ in streams for x in (
                self._stream_rows(O_buf, w, name, d), W16(f"{p}.attn.{outn}.weight"), W16(f"{p}.attn.{outn}.bias"),
                self._stream_rows(w.X, w, name, d))), d, d, EPI_BIAS_GATE_RES, gate1=mods["txt"][:, 2 * d:3 * d],
                gate2=mods["img"][:, 2 * d:3 * d], gate_ld=6 * d, aux1=part(aux1, "txt"), aux2=part(aux1, "img"))
            xm = keep["x_mid"] if keep is not None else (save["x_mid"] if save is not None else None)
            if xm is not None:
                xm.copy_(w.X)
        x_mid = keep["x_mid"] if replay else w.X                 # replay: saved, no to_out GEMM in the recompute
        nrm2 = w.nrm if save is None else save["nrm2"]
        for name in ("txt", "img"):
            m = mods[name]
            ops.ln_modulate(self._stream_rows(x_mid, w, name, d), m[:, 3 * d:4 * d], m[:, 4 * d:5 * d], 6 * d, nrm2[sl[name]], d)
        if replay and ff_kept:
            pass        # pre-activation kept: no GEMM, and no activation either -- its only reader in the backward, the weight
                        # gradient of ff.net.2, applies GELU inside its operand transpose (flux_backward._wgrad)
        else:
            ops.gemm_pair(*(x for name, _, _, _, _, _, ffn, *_ in streams for x in (
                Rows.of(nrm2[sl[name]]), W16(f"{p}.{ffn}.net.0.proj.weight"), W16(f"{p}.{ffn}.net.0.proj.bias"),
                Rows.of(w.hid[sl[name]]))), 4 * d, d, EPI_BIAS_GELU, aux1=part(hpre, "txt"), aux2=part(hpre, "img"))
        if not replay:                                           # y_ff saved: no ff.net.2 GEMM in the recompute
            ops.gemm_pair(*(x for name, _, _, _, _, _, ffn, *_ in streams for x in (
                Rows.of(w.hid[sl[name]]), W16(f"{p}.{ffn}.net.2.weight"), W16(f"{p}.{ffn}.net.2.bias"),
                self._stream_rows(w.X, w, name, d))), d, 4 * d, EPI_BIAS_GATE_RES, gate1=mods["txt"][:, 5 * d:6 * d],
                gate2=mods["img"][:, 5 * d:6 * d], gate_ld=6 * d, aux1=part(aux2, "txt"), aux2=part(aux2, "img"))
        return mods

    def _single_block(self, i, w, st, cos, sin, save=None, mod_in=None, keep=None, replay=False, x_in=None):
        """`keep` / `replay` / `x_in`: as in `_double_block`, with per-block buffers {O, lse, y_attn}."""
        cfg, d, H = self.cfg, self.cfg.dim, self.cfg.num_attention_heads
        p = f"single_transformer_blocks.{i}"
        B, S = w.B, w.S
        M = B * S
        if mod_in is not None:
            m = mod_in
        else:
            m = torch.empty(B, 3 * d, dtype=BF16, device=self.store.device)
            ops.skinny_linear(st, self.W(f"{p}.norm.linear.weight"), self.W(f"{p}.norm.linear.bias"), m, 3 * d, d)
        Xa = Rows(w.X if x_in is None else x_in, M, d, S, S * d)
        nrm = w.nrm if save is None else save["nrm1"]
        ops.ln_modulate(Xa, m[:, 0:d], m[:, d:2 * d], 3 * d, nrm, d)
        qkv_kept = keep is not None and "qkv" in keep           # as in `_double_block`
        qkv = keep["qkv"] if qkv_kept else w.qkv
        nograd = save is None and keep is None
        if nograd:
            self._qkv_nograd(nrm, f"{p}.attn.to_q", qkv, w, S, 0, self.W32(f"{p}.attn.norm_q.weight"),
                             self.W32(f"{p}.attn.norm_k.weight"), cos, sin)
        elif not (replay and qkv_kept):
            ops.gemm(Rows.of(nrm), self.store.fused(self.store.w16, f"{p}.attn.to_q.weight", 3 * d),
                     self.store.fused(self.store.w16, f"{p}.attn.to_q.bias", 3 * d), Rows.of(qkv), 3 * d, d)
        cat2 = w.cat.view(M, 5 * d)
        # the pre-activation (when kept) goes to columns 3d..7d of the [M, 7d] gradient staging buffer's twin
        ff_kept = keep is not None and "hid_pre" in keep
        if replay and ff_kept:                       # pre-activation kept by the forward (`save["hid_pre"]` IS that buffer):
            pass                                     # no GEMM and no [O | mlp] operand (flux_backward: `lean`)
        else:
            ops.gemm(Rows.of(nrm), self.W(f"{p}.proj_mlp.weight"), self.W(f"{p}.proj_mlp.bias"),
                     Rows(cat2[0, d:], M, 5 * d), 4 * d, d, EPI_BIAS_GELU,
                     aux=save["hid_pre"] if save is not None else (keep["hid_pre"] if ff_kept else None))
        if not nograd:
            ops.qk_norm_rope(qkv, self.W32(f"{p}.attn.norm_q.weight"), self.W32(f"{p}.attn.norm_k.weight"), cos, sin,
                             w.Q, w.K, w.Vt, B, H, S, w.Sp, S, 0, q_scale=self.q_scale(),
                             **({} if save is None else dict(V=save["V"], Qt=save["Qt"], Kt=save["Kt"])))
        if replay:                                   # attention output and proj_out result were kept by the forward
            if not ff_kept:
                w.cat[:, :, :d].copy_(keep["O"])     # (the wgrad of proj_out then reads [O | mlp] as one [M, 5d] operand)
            return m
        self._attn(w, w.cat, keep["lse"] if keep is not None else (w.lse if save is not None else None), 5 * d, S * 5 * d)
        if keep is not None:
            keep["O"].copy_(w.cat[:, :, :d])
        aux = keep["y_attn"] if keep is not None else (None if save is None else save["y_attn"])
        ops.gemm(Rows.of(cat2), self.W(f"{p}.proj_out.weight"), self.W(f"{p}.proj_out.bias"), Xa, d, 5 * d,
                 EPI_BIAS_GATE_RES, gate=m[:, 2 * d:3 * d], gate_ld=3 * d, aux=aux)
        return m

    def _embed(self, w, hidden_states, encoder_hidden_states):
        d = self.cfg.dim
        B, N, L = w.B, w.N, w.L
        hs = hidden_states.detach()
        if hs.dtype == BF16:
            w.in16.copy_(hs.reshape(B * N, -1))
        else:
            ops.cast_bf16(hs.to(F32).contiguous().view(-1), w.in16.view(-1))
        ops.gemm(Rows.of(w.in16), self.W("x_embedder.weight"), self.W("x_embedder.bias"),
                 self._stream_rows(w.X, w, "img", d), d, self.cfg.in_channels)
        ehs = encoder_hidden_states.detach().to(BF16).contiguous()
        ops.gemm(Rows.of(ehs.view(B * L, -1)), self.W("context_embedder.weight"), self.W("context_embedder.bias"),
                 self._stream_rows(w.X, w, "txt", d), d, self.cfg.joint_attention_dim)
        return ehs

    def _head(self, w, st):
        d = self.cfg.dim
        B, N = w.B, w.N
        e = torch.empty(B, 2 * d, dtype=BF16, device=self.store.device)
        ops.skinny_linear(st, self.W("norm_out.linear.weight"), self.W("norm_out.linear.bias"), e, 2 * d, d)
        nrm = w.nrm[:B * N]
        ops.ln_modulate(self._stream_rows(w.X, w, "img", d), e[:, d:2 * d], e[:, 0:d], 2 * d, nrm, d)   # scale first
        cout = self.cfg.patch_size * self.cfg.patch_size * self.cfg.in_channels
        out = torch.empty(B, N, cout, dtype=BF16, device=self.store.device)
        ops.gemm(Rows.of(nrm), self.W("proj_out.weight"), self.W("proj_out.bias"), Rows.of(out.view(B * N, cout)), cout, d)
        return out, e

    def forward(self, hidden_states, encoder_hidden_states, timestep, guidance, txt_ids, pooled_projections, img_ids,
                joint_attention_kwargs=None, return_dict=False):
        if torch.is_grad_enabled() and self.training and self.flat_param.requires_grad:
            from .flux_backward import FluxFunction
            out = FluxFunction.apply(self.flat_param, self, hidden_states, encoder_hidden_states, timestep, guidance,
                                     txt_ids, pooled_projections, img_ids)
            return (out,)
        with torch.no_grad():
            out = self._forward_nograd(hidden_states, encoder_hidden_states, timestep, guidance, txt_ids,
                                       pooled_projections, img_ids)
        return (out,)

    def _forward_nograd(self, hidden_states, encoder_hidden_states, timestep, guidance, txt_ids, pooled_projections,
                        img_ids, collect=None):
        B, N, _ = hidden_states.shape
        L = encoder_hidden_states.shape[1]
        w = self._workspace(B, L, N)
        self._embed(w, hidden_states, encoder_hidden_states)
        temb, st = self._temb(B, timestep.to(self.store.device), guidance, pooled_projections)
        cos, sin = self._rope(txt_ids, img_ids)
        if collect is not None:
            collect.update(temb=temb.clone(), x_embed=w.X[:, L:].clone(), ctx_embed=w.X[:, :L].clone())
        for i in range(self.cfg.num_layers):
            self._double_block(i, w, st, cos, sin)
            if collect is not None:
                collect[f"double{i}_h"] = w.X[:, L:].clone()
                collect[f"double{i}_c"] = w.X[:, :L].clone()
        for i in range(self.cfg.num_single_layers):
            self._single_block(i, w, st, cos, sin)
            if collect is not None:
                collect[f"single{i}_x"] = w.X.clone()
        out, _ = self._head(w, st)
        return out

    # ------------------------------------------------------------------ checkpoint format (reference checkpoint.py:65-88)
    def save_pretrained(self, save_dir):
        from safetensors.torch import save_file
        os.makedirs(save_dir, exist_ok=True)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        save_file(sd, os.path.join(save_dir, "diffusion_pytorch_model.safetensors"))
        cfg = dict(self.config)
        cfg.pop("dtype", None)
        with open(os.path.join(save_dir, "config.json"), "w") as f:
            json.dump(cfg, f, indent=4)

    @classmethod
    def from_pretrained(cls, path, device="cuda", subfolder=None, torch_dtype=None):
        """`FluxTransformer2DModel.from_pretrained(model_path, subfolder="transformer", torch_dtype=torch.float32)` as the
        reference calls it (train_grpo_flux.py:677-679): config.json + diffusers-named safetensors, single file or the
        sharded layout of the published FLUX.1-dev checkpoint (`...safetensors.index.json`).  Master weights are always
        fp32 (`torch_dtype` is accepted for the call signature)."""
        from safetensors.torch import load_file
        if subfolder and os.path.isdir(os.path.join(path, subfolder)):
            path = os.path.join(path, subfolder)
        with open(os.path.join(path, "config.json")) as f:
            raw = json.load(f)
        keys = {f.name for f in FluxConfig.__dataclass_fields__.values()}
        cfg = FluxConfig(**{k: (tuple(v) if k == "axes_dims_rope" else v) for k, v in raw.items() if k in keys})
        m = cls(cfg, device=device)
        index = os.path.join(path, "diffusion_pytorch_model.safetensors.index.json")
        if os.path.exists(index):
            with open(index) as f:
                shards = sorted(set(json.load(f)["weight_map"].values()))
            sd = {}
            for shard in shards:
                sd.update(load_file(os.path.join(path, shard)))
        else:
            sd = load_file(os.path.join(path, "diffusion_pytorch_model.safetensors"))
        m.load_state_dict(sd)
        return m
