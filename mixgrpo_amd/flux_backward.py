"""Backward pass of the HIP FLUX MMDiT with block-level activation recompute.

Replaces `loss.backward()` through diffusers' FluxTransformer2DModel under FSDP + full activation
checkpointing (reference fastvideo/train_grpo_flux.py:585, fastvideo/utils/fsdp_util.py:26-53): the forward
keeps only each block's input (bf16 [B, S, d]); the backward re-runs one block at a time with its
intermediates kept, then walks it in reverse.  Weight gradients accumulate in fp32 into the flat gradient
buffer (MGX_EPI_F32_ACC, beta = 1); activation gradients are bf16 like autograd's.

All arithmetic is in csrc/ kernels; this file only sequences them.
"""
import math

import torch

from . import ops
from .ops import BF16, F32, Rows, EPI_BIAS, EPI_BIAS_GATE_RES, EPI_DGELU, EPI_F32_ACC


import os

KEEP_ACTS = os.environ.get("MGX_KEEP_ACTS", "1") != "0"
# How many blocks ALSO keep their FF / proj_mlp pre-activation ([M, 4d] bf16 per block: 0.79 GB at micro-batch 7), so that the
# recompute pass re-creates the activation with an elementwise GELU instead of re-running the d -> 4d GEMM (1/3 of a block's
# linear FLOPs).  An integer = that many leading blocks from the start; "auto" (default) = none until the trainer, at the
# start of its SECOND train step (the first one has shown the step's real peak), hands out what the device still has free
# beyond `KEEP_FF_RESERVE_GIB` (`FluxTransformer2DModel.grow_ff_keep`).  Values kept are the very values a recompute produces.
KEEP_FF = os.environ.get("MGX_KEEP_FF_BLOCKS", "auto")
KEEP_FF_RESERVE_GIB = float(os.environ.get("MGX_KEEP_FF_RESERVE_GIB", "12"))
# Likewise the fused QKV projection's output ([M, 3d] bf16 per block: 0.085 GB per sample): the recompute pass then skips the
# d -> 3d GEMM as well (1/4 of a block's linear FLOPs) and only re-runs the elementwise QK-RMSNorm + RoPE on the kept rows.
# "auto": handed out together with the FF pre-activations, block by block (FF first), from the memory that is still free.
# With both kept for every block the recompute pass runs NO GEMM at all (micro-batch 4 at FLUX.1-dev 1024^2 on 288 GB).
KEEP_QKV = os.environ.get("MGX_KEEP_QKV_BLOCKS", "auto")


def _pad64(n):
    return (n + 63) // 64 * 64


class _Train:
    """Extra buffers of the training pass for one (L, N) shape at a batch CAPACITY B (smaller micro-batches use
    leading slices: `view`)."""

    def __init__(self, cfg, w, device):
        d, H, hd = cfg.dim, cfg.num_attention_heads, cfg.attention_head_dim
        B, S, Sp = w.B, w.S, w.Sp
        self.B, self.S = B, S
        M = B * S
        e = lambda *shape, dtype=BF16: torch.empty(*shape, dtype=dtype, device=device)
        z = lambda *shape, dtype=BF16: torch.zeros(*shape, dtype=dtype, device=device)
        nblk = cfg.num_layers + cfg.num_single_layers
        self.block_in = e(nblk, B, S, d)          # block inputs: the recompute pass starts from these
        self.x_final = e(B, S, d)
        # Selective activation saving (288 GB HBM: no parameter sharding, so there is room): the attention output + LSE
        # and the pre-gate outputs of the projections that END a residual branch are kept per block (4 x 28 MB per
        # sample and double block, 2 x per single block), so the recompute pass skips attention and those GEMMs:
        # 5/12 of a block's linear FLOPs and all of its attention FLOPs.  MGX_KEEP_ACTS=0 restores full recompute.
        self.keep = None
        if KEEP_ACTS:
            self.keep = []
            for b in range(nblk):
                k = dict(O=e(B, S, d), lse=e(B, H, S, dtype=F32), y_attn=e(M, d))
                if b < cfg.num_layers:
                    k.update(y_ff=e(M, d), x_mid=e(B, S, d))
                self.keep.append(k)
            self._ff_shape, self._ff_device = (M, 4 * d), device
            self._qkv_shape = (M, 3 * d)
            if KEEP_FF != "auto":
                self.grow_ff(int(KEEP_FF))
            if KEEP_QKV != "auto":
                self.grow_qkv(int(KEEP_QKV))
        self.save = dict(nrm1=e(M, d), nrm2=e(M, d), y_attn=e(M, d), y_ff=e(M, d), hid_pre=e(M, 4 * d),
                         x_mid=e(B, S, d), V=e(B, H, S, hd), Qt=z(B, H, hd, Sp), Kt=z(B, H, hd, Sp))
        self.dX = e(B, S, d)
        self.dO = e(B, S, 5 * d)                  # dO [B,S,d] view for double blocks, d(cat) [B,S,5d] for single
        self.dQ, self.dK, self.dV = e(B, H, S, hd), e(B, H, S, hd), e(B, H, S, hd)
        self.dOt = z(B, H, hd, Sp)
        self.delta = e(B, H, S, dtype=F32)
        self.dy = e(M, d)
        self.dbig = e(M, 7 * d)                   # [dqkv | dmlp_pre] (single) / dqkv, dhid_pre (double)
        self.dnrm = e(M, d)
        Mp = _pad64(M)
        self.dCt = e(7 * d * Mp)                  # transposed output-gradient operand for wgrad
        self.At = e(5 * d * Mp)                   # transposed activation operand for wgrad
        self.Wt = e(7 * d * max(d, 64) + 5 * d * d)
        self.ones = torch.ones(1, 5 * d, dtype=BF16, device=device)

    def view(self, B):
        return self if B == self.B else _TrainView(self, B)

    def ff_block_bytes(self):
        return self._ff_shape[0] * self._ff_shape[1] * 2

    def ff_kept(self):
        return sum(1 for k in (self.keep or []) if "hid_pre" in k)

    def qkv_block_bytes(self):
        return self._qkv_shape[0] * self._qkv_shape[1] * 2

    def qkv_kept(self):
        return sum(1 for k in (self.keep or []) if "qkv" in k)

    def grow_ff(self, n_total):
        """Keep the FF pre-activation of the first `n_total` blocks (allocates the missing buffers)."""
        if self.keep is None:
            return 0
        for k in self.keep[:max(0, n_total)]:
            if "hid_pre" not in k:
                k["hid_pre"] = torch.empty(*self._ff_shape, dtype=BF16, device=self._ff_device)
        return self.ff_kept()

    def grow_qkv(self, n_total):
        """Keep the QKV projection's output of the first `n_total` blocks (allocates the missing buffers)."""
        if self.keep is None:
            return 0
        for k in self.keep[:max(0, n_total)]:
            if "qkv" not in k:
                k["qkv"] = torch.empty(*self._qkv_shape, dtype=BF16, device=self._ff_device)
        return self.qkv_kept()

    def grow_auto(self, budget_bytes):
        """Spend `budget_bytes` block by block, FF pre-activation before QKV output, on whatever is still recomputed and
        set to "auto".  Returns the bytes left."""
        if self.keep is None:
            return budget_bytes
        for i, k in enumerate(self.keep):
            for name, auto, size, grow in (("hid_pre", KEEP_FF == "auto", self.ff_block_bytes(), self.grow_ff),
                                           ("qkv", KEEP_QKV == "auto", self.qkv_block_bytes(), self.grow_qkv)):
                if name in k or not auto:
                    continue
                if budget_bytes < size:
                    return budget_bytes
                k[name] = torch.empty(*(self._ff_shape if name == "hid_pre" else self._qkv_shape), dtype=BF16,
                                      device=self._ff_device)
                budget_bytes -= size
        return budget_bytes


class _TrainView:
    """The first B batches of a `_Train` (batch-major contiguous buffers: plain prefixes; flat operand buffers shared)."""

    def __init__(self, base, B):
        assert B <= base.B
        M = B * base.S
        self.B, self.S = B, base.S
        self.block_in = [base.block_in[i, :B] for i in range(base.block_in.shape[0])]
        self.x_final = base.x_final[:B]
        rows = ("nrm1", "nrm2", "y_attn", "y_ff", "hid_pre")
        self.save = {k: (v[:M] if k in rows else v[:B]) for k, v in base.save.items()}
        self.keep = None
        if base.keep is not None:
            self.keep = [{k: (v[:M] if k in ("y_attn", "y_ff", "hid_pre", "qkv") else v[:B]) for k, v in kb.items()}
                         for kb in base.keep]
        self.dX, self.dO, self.dQ, self.dK, self.dV, self.dOt, self.delta = (
            t[:B] for t in (base.dX, base.dO, base.dQ, base.dK, base.dV, base.dOt, base.delta))
        self.dy, self.dbig, self.dnrm = base.dy[:M], base.dbig[:M], base.dnrm[:M]
        self.dCt, self.At, self.Wt, self.ones = base.dCt, base.At, base.Wt, base.ones


def _train_buffers(cfg, w, device):
    """`_Train` view for the workspace view `w`; the base grows to the largest training batch seen."""
    base = w.train
    if base is None or base.B < w.B:
        w.train = None
        del base
        torch.cuda.empty_cache()
        base = _Train(cfg, w, device)
        w.train = base
    return base.view(w.B)


class FluxFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flat_param, model, hidden_states, encoder_hidden_states, timestep, guidance, txt_ids,
                pooled_projections, img_ids):
        cfg = model.cfg
        B, N, _ = hidden_states.shape
        L = encoder_hidden_states.shape[1]
        w = model._workspace(B, L, N)
        tr = _train_buffers(cfg, w, model.store.device)
        ehs = model._embed(w, hidden_states, encoder_hidden_states)
        keep = {}
        temb, st = model._temb(B, timestep.to(model.store.device), guidance, pooled_projections, keep=keep)
        cos, sin = model._rope(txt_ids, img_ids)
        mods = []
        blk = 0
        for i in range(cfg.num_layers):
            tr.block_in[blk].copy_(w.X)
            mods.append(model._double_block(i, w, st, cos, sin, keep=tr.keep[blk] if tr.keep else None))
            blk += 1
        for i in range(cfg.num_single_layers):
            tr.block_in[blk].copy_(w.X)
            mods.append(model._single_block(i, w, st, cos, sin, keep=tr.keep[blk] if tr.keep else None))
            blk += 1
        tr.x_final.copy_(w.X)
        out, e = model._head(w, st)
        # the saved activations live in the model's shared per-shape workspace: ONE pending backward per model at a time
        # (INTEGRATION.md).  A second grad-enabled forward before this one's backward would overwrite them: stamped here,
        # checked in backward.
        w.train.generation = getattr(w.train, "generation", 0) + 1
        ctx.generation = w.train.generation
        ctx.model, ctx.w, ctx.tr = model, w, tr
        ctx.saved = dict(ehs=ehs, in16=w.in16.clone(), temb=temb, st=st, keep=keep, cos=cos, sin=sin, mods=mods, e=e)
        return out

    @staticmethod
    def backward(ctx, dout):
        if getattr(ctx.w.train, "generation", None) != ctx.generation:
            from ._lib import MgxError
            raise MgxError("FluxTransformer2DModel: the activations of this forward pass were overwritten by a later "
                           "grad-enabled forward of the same model (one pending backward per model; call backward() "
                           "before the next training forward)")
        _backward(ctx.model, ctx.w, ctx.tr, ctx.saved, dout.contiguous().to(BF16))
        return (None,) * 9


# ------------------------------------------------------------------------------------------------ helpers
def _wgrad(model, tr, A, K, dC: Rows, N, wname, bname, rows=None):
    """g32[W] += dC^T A ; g32[b] += colsum(dC).  `rows`: number of fused weight rows when W spans several tensors.
    `A`: the [M, K] activation as `Rows`, or a list of column blocks `(Rows, width, gelu)` that make it up side by side --
    `gelu`: that block is gelu_tanh of the given (kept) pre-activation, applied on the way into the transposed operand."""
    st = model.store
    g = st.ensure_grad()
    parts = A if isinstance(A, list) else [(A, K, False)]
    M = parts[0][0].M
    Mp = _pad64(M)
    dCt = tr.dCt[:N * Mp].view(N, Mp)
    At = tr.At[:K * Mp].view(K, Mp)
    gb = None
    if bname is not None:
        gb = st.fused(g, bname, N) if rows else st.view(g, bname)
    ops.transpose(dC, N, dCt, Mp, colsum_out=gb, colsum_beta=1.0)
    k0 = 0
    for a_rows, width, gelu in parts:
        ops.transpose(a_rows, width, At[k0:k0 + width], Mp, gelu=gelu)
        k0 += width
    assert k0 == K
    gw = st.fused(g, wname, N) if rows else st.view(g, wname)
    ops.gemm(Rows.of(dCt), At, None, Rows(gw, N, K), K, Mp, EPI_F32_ACC, beta=1.0, ldw=Mp)


def _dgrad(model, tr, dC: Rows, N, K, wname, out: Rows, rows=None, epi=EPI_BIAS, aux=None, ldaux=None, row_lo=0,
           row_hi=None, gate=None):
    """out[M, K'] = epi(dC[M, N] @ W[N, K][:, row_lo:row_hi]) via a transposed copy of W (K-contiguous operand)."""
    st = model.store
    W = st.fused(st.w16, wname, N) if rows else st.view(st.w16, wname)
    Wt = tr.Wt[:K * N].view(K, N)
    ops.transpose(Rows.of(W), K, Wt, N)
    row_hi = K if row_hi is None else row_hi
    ops.gemm(dC, Wt[row_lo:row_hi], None, out, row_hi - row_lo, N, epi, aux=aux, ldaux=ldaux, gate=gate, gate_ld=0)


def _dgrad_pair(model, tr, dC1: Rows, wname1, out1: Rows, dC2: Rows, wname2, out2: Rows, N, K, rows=None, epi=EPI_BIAS, aux1=None,
                aux2=None, ldaux=None):
    """`_dgrad` for the text- and the image-stream Linear of a double block in ONE launch (`ops.gemm_pair`): both transposed
    weights side by side in the Wt scratch, problem 1 = the text stream."""
    st = model.store
    Wts = []
    for j, wname in enumerate((wname1, wname2)):
        W = st.fused(st.w16, wname, N) if rows else st.view(st.w16, wname)
        Wt = tr.Wt[j * K * N:(j + 1) * K * N].view(K, N)
        ops.transpose(Rows.of(W), K, Wt, N)
        Wts.append(Wt)
    ops.gemm_pair(dC1, Wts[0], None, out1, dC2, Wts[1], None, out2, K, N, epi, aux1=aux1, aux2=aux2, ldaux=ldaux)


def _skinny_bwd(model, tr, dmod, x, wname, bname, N, K, dx_acc):
    """Modulation / embedder linear backward: g32[W] += dmod^T x, g32[b] += sum_b dmod, dx_acc += dmod @ W."""
    st = model.store
    g = st.ensure_grad()
    ops.skinny_wgrad(dmod, x, st.view(g, wname), st.view(g, bname), N, K)
    if dx_acc is not None:
        ops.skinny_dgrad(dmod, st.view(st.w16, wname), dx_acc, N, K)      # straight from the row-major weight


def _backward(model, w, tr, sv, dout):
    cfg = model.cfg
    d, H = cfg.dim, cfg.num_attention_heads
    B, L, N, S, Sp = w.B, w.L, w.N, w.S, w.Sp
    M = B * S
    dev = model.store.device
    st_, cos, sin = sv["st"], sv["cos"], sv["sin"]
    scale, q_scale = model.attn_scale(), model.q_scale()   # (ln 2, scale * log2 e) for the prescaled Q of mgx_attn_fwd_log2
    store = model.store
    g32 = store.ensure_grad()
    dst = torch.zeros(B, d, dtype=BF16, device=dev)        # grad wrt st = silu(temb), summed over all users
    save0 = tr.save
    save = save0
    row0 = {"txt": 0, "img": B * L}

    def srows(t, which, width):
        return model._stream_rows(t, w, which, width)

    # ---------------- head: out = proj_out( LN(x_img) * (1+scale) + shift )
    cout = cfg.patch_size * cfg.patch_size * cfg.in_channels
    e = sv["e"]
    nrm = save["nrm1"][:B * N]
    ops.ln_modulate(srows(tr.x_final, "img", d), e[:, d:2 * d], e[:, 0:d], 2 * d, nrm, d)
    dC = Rows.of(dout.view(B * N, cout))
    _wgrad(model, tr, Rows.of(nrm), d, dC, cout, "proj_out.weight", "proj_out.bias")
    dn = tr.dnrm[:B * N]
    _dgrad(model, tr, dC, cout, d, "proj_out.weight", Rows.of(dn))
    tr.dX.zero_()
    de = torch.zeros(B, 2 * d, dtype=BF16, device=dev)
    ops.ln_modulate_bwd(dn, srows(tr.x_final, "img", d), e[:, 0:d], 2 * d, srows(tr.dX, "img", d), False,
                        de[:, d:2 * d], de[:, 0:d], d)
    _skinny_bwd(model, tr, de, st_, "norm_out.linear.weight", "norm_out.linear.bias", 2 * d, d, dst)

    # data-parallel overlap (dist_utils.GradReducer): set by the trainer for the LAST micro-batch before an optimizer step;
    # called with a block's prefix once every gradient of that block is final
    grad_ready = getattr(model, "_grad_ready", None) or (lambda prefix: None)
    grad_ready("tail")
    nblk = cfg.num_layers + cfg.num_single_layers
    # ---------------- single blocks (reverse)
    for i in reversed(range(cfg.num_single_layers)):
        blk = cfg.num_layers + i
        p = f"single_transformer_blocks.{i}"
        m = sv["mods"][blk]
        kept = tr.keep[blk] if tr.keep else None
        save = dict(save0, y_attn=kept["y_attn"]) if kept else save0
        if kept and "hid_pre" in kept:
            save["hid_pre"] = kept["hid_pre"]                   # kept by the forward: the replay skips proj_mlp
        if kept is None:
            w.X.copy_(tr.block_in[blk])                        # full recompute rewrites the residual stream
        model._single_block(i, w, st_, cos, sin, save=save, mod_in=m, keep=kept, replay=kept is not None,
                            x_in=tr.block_in[blk] if kept is not None else None)
        lse_b = kept["lse"] if kept is not None else w.lse
        dmod = torch.empty(B, 3 * d, dtype=BF16, device=dev)
        x_in = Rows(tr.block_in[blk], M, d, S, S * d)
        dXr = Rows(tr.dX, M, d, S, S * d)
        cat2 = w.cat.view(M, 5 * d)
        # everything kept (attention output AND the FF pre-activation): the recompute pass built no [O | mlp] operand -- the
        # weight gradient of proj_out takes O from its keep buffer and gelu(hid_pre) formed inside the transpose, and the
        # attention backward reads O / writes dO at row stride d
        lean = kept is not None and "hid_pre" in kept
        # out = x + gate * y ; y = proj_out(cat)
        ops.gate_bwd(dXr, save["y_attn"], m[:, 2 * d:3 * d], 3 * d, tr.dy, dmod[:, 2 * d:3 * d], B, S, d)
        dyr = Rows.of(tr.dy)
        a_op = [(Rows.of(kept["O"].view(M, d)), d, False), (Rows.of(kept["hid_pre"]), 4 * d, True)] if lean else Rows.of(cat2)
        _wgrad(model, tr, a_op, 5 * d, dyr, d, f"{p}.proj_out.weight", f"{p}.proj_out.bias")
        dcat = tr.dO.view(M, 5 * d)
        dO1 = tr.dO.view(-1)[:M * d].view(B, S, d)             # lean: dO alone, [B, S, d]
        dbig = tr.dbig                                         # [M, 7d] = [dq | dk | dv | dmlp_pre]
        # d(cat)[:, :d] = dO (attention output grad) ; d(cat)[:, d:] -> through GELU -> dbig[:, 3d:]
        stW = store.view(store.w16, f"{p}.proj_out.weight")
        Wt = tr.Wt[:5 * d * d].view(5 * d, d)
        ops.transpose(Rows.of(stW), 5 * d, Wt, d)
        ops.gemm(dyr, Wt[0:d], None, Rows.of(dO1.view(M, d)) if lean else Rows(dcat, M, 5 * d), d, d, EPI_BIAS)
        ops.gemm(dyr, Wt[d:5 * d], None, Rows(dbig[0, 3 * d:], M, 7 * d), 4 * d, d, EPI_DGELU, aux=save["hid_pre"],
                 ldaux=4 * d)
        if lean:
            ops.attn_bwd(w.Q, w.K, save["V"], save["Qt"], save["Kt"], kept["O"], dO1, lse_b, tr.delta, tr.dOt, tr.dQ, tr.dK,
                         tr.dV, B, H, S, Sp, d, S * d, scale)
        else:
            ops.attn_bwd(w.Q, w.K, save["V"], save["Qt"], save["Kt"], w.cat, tr.dO, lse_b, tr.delta, tr.dOt, tr.dQ, tr.dK,
                         tr.dV, B, H, S, Sp, 5 * d, S * 5 * d, scale)
        # qk norm / rope backward writes [dq|dk|dv] straight into columns 0..3d of the [M, 7d] staging matrix
        qkv_b = kept["qkv"] if kept is not None and "qkv" in kept else w.qkv
        ops.qk_norm_rope_bwd(qkv_b, model.W32(f"{p}.attn.norm_q.weight"), model.W32(f"{p}.attn.norm_k.weight"), cos, sin,
                             tr.dQ, tr.dK, tr.dV, dbig, store.view(g32, f"{p}.attn.norm_q.weight"),
                             store.view(g32, f"{p}.attn.norm_k.weight"), B, H, S, Sp, S, 0, ld_dqkv=7 * d, q_scale=q_scale)
        dbr = Rows.of(dbig)
        _wgrad(model, tr, Rows.of(save["nrm1"]), d, dbr, 7 * d, f"{p}.attn.to_q.weight", f"{p}.attn.to_q.bias", rows=True)
        _dgrad(model, tr, dbr, 7 * d, d, f"{p}.attn.to_q.weight", Rows.of(tr.dnrm), rows=True)
        ops.ln_modulate_bwd(tr.dnrm, x_in, m[:, d:2 * d], 3 * d, dXr, True, dmod[:, 0:d], dmod[:, d:2 * d], d)
        _skinny_bwd(model, tr, dmod, st_, f"{p}.norm.linear.weight", f"{p}.norm.linear.bias", 3 * d, d, dst)
        grad_ready(p)

    # ---------------- double blocks (reverse)
    # text stream first = problem 1 of the pair launches (its rows come first in every stacked buffer); the two streams'
    # input-gradient GEMMs go out as one launch each, everything else per stream
    streams = (("txt", "norm1_context", ("add_q_proj", "add_k_proj", "add_v_proj"), "norm_added_q", "norm_added_k",
                "to_add_out", "ff_context", L, 0),
               ("img", "norm1", ("to_q", "to_k", "to_v"), "norm_q", "norm_k", "to_out.0", "ff", N, L))
    sl = {"txt": slice(0, B * L), "img": slice(B * L, B * S)}
    dO3 = tr.dO.view(-1)[:B * S * d].view(B, S, d)
    dh_all = tr.dbig.view(-1)[:M * 4 * d].view(M, 4 * d)
    dqkv_all = tr.dbig.view(-1)[:M * 3 * d].view(M, 3 * d)
    for i in reversed(range(cfg.num_layers)):
        p = f"transformer_blocks.{i}"
        mods = sv["mods"][i]
        kept = tr.keep[i] if tr.keep else None
        save = dict(save0, y_attn=kept["y_attn"], y_ff=kept["y_ff"], x_mid=kept["x_mid"]) if kept else save0
        if kept and "hid_pre" in kept:
            save["hid_pre"] = kept["hid_pre"]                   # kept by the forward: the replay skips ff.net.0
        if kept is None:
            w.X.copy_(tr.block_in[i])                          # full recompute rewrites the residual stream
        model._double_block(i, w, st_, cos, sin, save=save, mods_in=mods, keep=kept, replay=kept is not None,
                            x_in=tr.block_in[i] if kept is not None else None)
        O_b, lse_b = (kept["O"], kept["lse"]) if kept is not None else (w.O, w.lse)
        dmods = {k: torch.empty(B, 6 * d, dtype=BF16, device=dev) for k in ("img", "txt")}
        # ---- FF branch: out = x_mid + gate_mlp * ff2(gelu(ff1(LNmod(x_mid))))
        for name, norm, qkvn, nq, nk, outn, ffn, rows, s0 in streams:
            m, dm = mods[name], dmods[name]
            ops.gate_bwd(srows(tr.dX, name, d), save["y_ff"][sl[name]], m[:, 5 * d:6 * d], 6 * d, tr.dy[sl[name]],
                         dm[:, 5 * d:6 * d], B, rows, d)
            # (FF pre-activation kept: the recompute pass formed no activation; gelu(hid_pre) is applied inside the transpose)
            a_op = [(Rows.of(kept["hid_pre"][sl[name]]), 4 * d, True)] if kept is not None and "hid_pre" in kept \
                else Rows.of(w.hid[sl[name]])
            _wgrad(model, tr, a_op, 4 * d, Rows.of(tr.dy[sl[name]]), d, f"{p}.{ffn}.net.2.weight", f"{p}.{ffn}.net.2.bias")
        _dgrad_pair(model, tr, Rows.of(tr.dy[sl["txt"]]), f"{p}.ff_context.net.2.weight", Rows.of(dh_all[sl["txt"]]),
                    Rows.of(tr.dy[sl["img"]]), f"{p}.ff.net.2.weight", Rows.of(dh_all[sl["img"]]), d, 4 * d, epi=EPI_DGELU,
                    aux1=save["hid_pre"][sl["txt"]], aux2=save["hid_pre"][sl["img"]], ldaux=4 * d)
        for name, norm, qkvn, nq, nk, outn, ffn, rows, s0 in streams:
            _wgrad(model, tr, Rows.of(save["nrm2"][sl[name]]), d, Rows.of(dh_all[sl[name]]), 4 * d, f"{p}.{ffn}.net.0.proj.weight",
                   f"{p}.{ffn}.net.0.proj.bias")
        _dgrad_pair(model, tr, Rows.of(dh_all[sl["txt"]]), f"{p}.ff_context.net.0.proj.weight", Rows.of(tr.dnrm[sl["txt"]]),
                    Rows.of(dh_all[sl["img"]]), f"{p}.ff.net.0.proj.weight", Rows.of(tr.dnrm[sl["img"]]), 4 * d, d)
        for name, norm, qkvn, nq, nk, outn, ffn, rows, s0 in streams:
            m, dm = mods[name], dmods[name]
            dXs = srows(tr.dX, name, d)
            ops.ln_modulate_bwd(tr.dnrm[sl[name]], srows(save["x_mid"], name, d), m[:, 4 * d:5 * d], 6 * d, dXs, True,
                                dm[:, 3 * d:4 * d], dm[:, 4 * d:5 * d], d)
            # ---- attention branch: x_mid = x_in + gate_msa * to_out(O)
            ops.gate_bwd(dXs, save["y_attn"][sl[name]], m[:, 2 * d:3 * d], 6 * d, tr.dy[sl[name]], dm[:, 2 * d:3 * d], B, rows, d)
            _wgrad(model, tr, srows(O_b, name, d), d, Rows.of(tr.dy[sl[name]]), d, f"{p}.attn.{outn}.weight",
                   f"{p}.attn.{outn}.bias")
        _dgrad_pair(model, tr, Rows.of(tr.dy[sl["txt"]]), f"{p}.attn.to_add_out.weight", srows(dO3, "txt", d),
                    Rows.of(tr.dy[sl["img"]]), f"{p}.attn.to_out.0.weight", srows(dO3, "img", d), d, d)
        ops.attn_bwd(w.Q, w.K, save["V"], save["Qt"], save["Kt"], O_b, dO3, lse_b, tr.delta, tr.dOt, tr.dQ, tr.dK, tr.dV,
                     B, H, S, Sp, d, S * d, scale)
        qkv_b = kept["qkv"] if kept is not None and "qkv" in kept else w.qkv
        for name, norm, qkvn, nq, nk, outn, ffn, rows, s0 in streams:
            ops.qk_norm_rope_bwd(qkv_b[sl[name]], model.W32(f"{p}.attn.{nq}.weight"), model.W32(f"{p}.attn.{nk}.weight"), cos,
                                 sin, tr.dQ, tr.dK, tr.dV, dqkv_all[sl[name]], store.view(g32, f"{p}.attn.{nq}.weight"),
                                 store.view(g32, f"{p}.attn.{nk}.weight"), B, H, S, Sp, rows, s0, q_scale=q_scale)
            _wgrad(model, tr, Rows.of(save["nrm1"][sl[name]]), d, Rows.of(dqkv_all[sl[name]]), 3 * d, f"{p}.attn.{qkvn[0]}.weight",
                   f"{p}.attn.{qkvn[0]}.bias", rows=True)
        _dgrad_pair(model, tr, Rows.of(dqkv_all[sl["txt"]]), f"{p}.attn.add_q_proj.weight", Rows.of(tr.dnrm[sl["txt"]]),
                    Rows.of(dqkv_all[sl["img"]]), f"{p}.attn.to_q.weight", Rows.of(tr.dnrm[sl["img"]]), 3 * d, d, rows=True)
        for name, norm, qkvn, nq, nk, outn, ffn, rows, s0 in streams:
            m, dm = mods[name], dmods[name]
            ops.ln_modulate_bwd(tr.dnrm[sl[name]], srows(tr.block_in[i], name, d), m[:, d:2 * d], 6 * d, srows(tr.dX, name, d),
                                True, dm[:, 0:d], dm[:, d:2 * d], d)
            _skinny_bwd(model, tr, dm, st_, f"{p}.{norm}.linear.weight", f"{p}.{norm}.linear.bias", 6 * d, d, dst)
        grad_ready(p)

    # ---------------- embedders (inputs need no gradient)
    _wgrad(model, tr, Rows.of(sv["in16"]), cfg.in_channels, srows(tr.dX, "img", d), d, "x_embedder.weight",
           "x_embedder.bias")
    _wgrad(model, tr, Rows.of(sv["ehs"].view(B * L, -1)), cfg.joint_attention_dim, srows(tr.dX, "txt", d), d,
           "context_embedder.weight", "context_embedder.bias")

    # ---------------- temb path: st = silu(temb); temb = sum of three MLP outputs
    dtemb = torch.empty(B, d, dtype=BF16, device=dev)
    ops.ew(sv["temb"], dst, dtemb, 1)
    for name, (x, h1, a1) in sv["keep"].items():
        K = x.shape[1]
        da1 = torch.zeros(B, d, dtype=BF16, device=dev)
        _skinny_bwd(model, tr, dtemb, a1, f"time_text_embed.{name}.linear_2.weight",
                    f"time_text_embed.{name}.linear_2.bias", d, d, da1)
        dh1 = torch.empty(B, d, dtype=BF16, device=dev)
        ops.ew(h1, da1, dh1, 1)
        _skinny_bwd(model, tr, dh1, x, f"time_text_embed.{name}.linear_1.weight",
                    f"time_text_embed.{name}.linear_1.bias", d, K, None)
    if model.flat_param.grad is None:
        model.flat_param.grad = g32
