"""Thin Python wrappers over the C-ABI MMDiT kernels (device tensors in, device tensors out, current stream).
Only plumbing lives here: shape bookkeeping and pointer passing.  See include/mixgrpo_hip.h for semantics."""
import os
from dataclasses import dataclass

import torch

from ._lib import check, lib, ptr, stream

EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_GATE_RES, EPI_F32_ACC, EPI_DGELU = 0, 1, 2, 3, 4
BF16 = torch.bfloat16
F32 = torch.float32


@dataclass
class Rows:
    """A row-batched bf16/fp32 matrix view: row m at data_ptr + (m // rpb) * bstride + (m % rpb) * ld elements."""
    t: torch.Tensor      # tensor whose data_ptr() is the first element
    M: int               # number of rows
    ld: int
    rpb: int = 1 << 40
    bstride: int = 0

    @staticmethod
    def of(t):
        """Plain 2-D view of a contiguous tensor [..., cols]."""
        cols = t.shape[-1]
        return Rows(t, t.numel() // cols, cols)


def gemm(A: Rows, W, bias, C: Rows, N, K, epi=EPI_BIAS, gate=None, gate_ld=0, aux=None, beta=0.0, ldw=None, ldaux=None):
    """C = epi(A @ W[N,K]^T + bias).  `aux`: plain [M, ldaux] matrix (ldaux defaults to N); pass a tensor whose
    data_ptr() is its first element."""
    assert A.M == C.M
    if GEMM_PROFILE is not None:     # bench.py: HIP events around every GEMM launch on the launch stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _gemm_launch(A, W, bias, C, N, K, epi, gate, gate_ld, aux, beta, ldw, ldaux)
        e1.record()
        GEMM_PROFILE.append((e0, e1, 2.0 * A.M * N * K, (A.M, N, K, epi)))
        return
    _gemm_launch(A, W, bias, C, N, K, epi, gate, gate_ld, aux, beta, ldw, ldaux)


GEMM_PROFILE = None

# MGX_LINEAR_VT=0: the value projection always leaves row-major and mgx_qk_norm_rope_fwd transposes it (the training passes
# do that anyway: their backward needs V row-major as well)
LINEAR_VT = os.environ.get("MGX_LINEAR_VT", "1") != "0"
# MGX_LINEAR_QKNORM=0: the q | k projection always writes its [tokens, 2d] output and mgx_qk_norm_rope_fwd_qs makes Q, K of it
LINEAR_QKNORM = os.environ.get("MGX_LINEAR_QKNORM", "1") != "0"
ROPE_PAIR_TABLE = os.environ.get("MGX_ROPE_PAIR_TABLE", "1") != "0"   # that epilogue reads (cos, sin) per pair when the tables allow


def linear_t(X, W, bias, Ct, tokens, F, K, ld_ct, tok_rpb, ct_bstride):
    """Ct[b][f][t] = bf16(X[b * tok_rpb + t] . W[f] + bias[f]) (`mgx_linear_bf16_t`: a Linear whose output leaves transposed,
    token-contiguous).  X plain [tokens, K], W [F, K]; Ct: a tensor whose data_ptr() is the element of (b 0, f 0, t 0).
    False -- nothing launched -- when the persistent kernel cannot take the shape."""
    ws = _sk_workspace(X.device) if GEMM_STREAM_K else None
    prof = GEMM_PROFILE is not None
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = lib().mgx_linear_bf16_t(ptr(X), ptr(W), ptr(bias), Ct.data_ptr(), tokens, F, K, K, K, ld_ct, tok_rpb, ct_bstride,
                                 None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel(), stream())
    if rc == 1:
        return False
    check(rc)
    if prof:
        e1.record()
        GEMM_PROFILE.append((e0, e1, 2.0 * tokens * F * K, (F, tokens, K, EPI_BIAS)))
    return True


def _gemm_launch(A, W, bias, C, N, K, epi, gate, gate_ld, aux, beta, ldw, ldaux):
    ws = _sk_workspace(C.t.device) if GEMM_STREAM_K else None
    check(lib().mgx_gemm_bf16_sk(ptr(A.t), ptr(W), ptr(bias), ptr(C.t), None if gate is None else gate.data_ptr(),
                                 None if aux is None else aux.data_ptr(), (N if ldaux is None else ldaux), A.M, N, K, A.ld,
                                 A.rpb, A.bstride, K if ldw is None else ldw, C.ld, C.rpb, C.bstride, gate_ld, epi, beta,
                                 None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel(), stream()))


def gemm_pair(A1: Rows, W1, bias1, C1: Rows, A2: Rows, W2, bias2, C2: Rows, N, K, epi=EPI_BIAS, gate1=None, gate2=None, gate_ld=0,
              aux1=None, aux2=None, ldaux=None):
    """Two Linears of equal shape in one launch (`mgx_gemm_bf16_pair`): problem 1 = the text stream, problem 2 = the image stream
    of a FLUX double block.  Operands as in `gemm`; A1 / A2 (and W1 / W2) must come from the same buffer, problem 1 first."""
    assert A1.M == C1.M and A2.M == C2.M and A1.ld == A2.ld and C1.ld == C2.ld
    if not GEMM_PAIR:
        gemm(A1, W1, bias1, C1, N, K, epi, gate=gate1, gate_ld=gate_ld, aux=aux1, ldaux=ldaux)
        gemm(A2, W2, bias2, C2, N, K, epi, gate=gate2, gate_ld=gate_ld, aux=aux2, ldaux=ldaux)
        return
    prof = GEMM_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    ws = _sk_workspace(C1.t.device) if GEMM_STREAM_K else None
    dp = lambda t: None if t is None else t.data_ptr()
    check(lib().mgx_gemm_bf16_pair(ptr(A1.t), ptr(W1), ptr(bias1), ptr(C1.t), dp(gate1), dp(aux1), A1.M, A1.rpb, A1.bstride, C1.rpb,
                                   C1.bstride, ptr(A2.t), ptr(W2), ptr(bias2), ptr(C2.t), dp(gate2), dp(aux2), A2.M, A2.rpb,
                                   A2.bstride, C2.rpb, C2.bstride, N, K, A1.ld, K, C1.ld, (N if ldaux is None else ldaux), gate_ld,
                                   epi, 0.0, dp(ws), 0 if ws is None else ws.numel(), stream()))
    if prof is not None:
        e1.record()
        prof.append((e0, e1, 2.0 * (A1.M + A2.M) * N * K, (A1.M + A2.M, N, K, epi)))


# MGX_GEMM_PAIR=1: the two streams' Linears of a double block go out as ONE launch (`mgx_gemm_bf16_pair`).  Off by default:
# measured neutral (same box, round 4: 18.418 s per step paired vs 18.397 s unpaired, 9745 vs 12861 launches, GEMM family 1374.5
# vs 1372.7 TFLOP/s; profiles/r04_bench_c_pair{1,0}.json.log).  The text launches' low rates (440-1240 TFLOP/s) are the idle
# capacity of a partial round, and the merged launch ends in the same partial round: the rollout's N = 3072 pair is 192 + 1536
# tiles = 0.75 + 6 rounds apart and 6.75 -> 7 rounds together.
GEMM_PAIR = os.environ.get("MGX_GEMM_PAIR", "0") == "1"


# Stream-K tail of the persistent GEMM (csrc/gemm.hip): on by default; MGX_GEMM_STREAM_K=0 calls the kernel without a
# workspace (every tile computed whole: results independent of the batch size).
GEMM_STREAM_K = os.environ.get("MGX_GEMM_STREAM_K", "1") != "0"
_sk_ws = {}


def _sk_workspace(device):
    """The caller-owned fp32 workspace of mgx_gemm_bf16_sk: one per (device, stream) that launches GEMMs."""
    key = (device, torch.cuda.current_stream().cuda_stream)
    w = _sk_ws.get(key)
    if w is None:
        w = torch.empty(lib().mgx_gemm_sk_workspace_elems(), dtype=F32, device=device)
        _sk_ws[key] = w
    return w


_scratch = {}


def scratch(name, numel, dtype, device):
    """Grow-only named scratch buffers (caller-owned workspace for the C ABI)."""
    key = (name, dtype, device)
    t = _scratch.get(key)
    if t is None or t.numel() < numel:
        t = torch.empty(int(numel), dtype=dtype, device=device)
        _scratch[key] = t
    return t[:numel]


def transpose(inp: Rows, N, out, ld_out, colsum_out=None, colsum_beta=1.0, gelu=False):
    """out[N, ld_out] = inp[M, N]^T (zero padded); optional colsum_out[n] = beta*colsum_out[n] + sum_m inp[m,n].
    `gelu`: out = gelu_tanh(inp)^T (plain matrix, no column sums): the activation of a kept FF pre-activation."""
    if gelu:
        assert colsum_out is None and inp.rpb >= inp.M
        check(lib().mgx_transpose_gelu_bf16(ptr(inp.t), ptr(out), inp.M, N, inp.ld, ld_out, stream()))
        return
    part = None
    if colsum_out is not None:
        part = scratch("colsum_partial", lib().mgx_transpose_partial_elems(inp.M, N), F32, out.device)
    check(lib().mgx_transpose_bf16(ptr(inp.t), ptr(out), ptr(part), ptr(colsum_out), colsum_beta, inp.M, N, inp.ld,
                                   inp.rpb, inp.bstride, ld_out, stream()))


def ln_modulate(x: Rows, shift, scale, mod_ld, y, D, stats=None):
    """y[M, D] = bf16(LN(x) * bf16(1+scale[b]) + shift[b]); shift/scale are views at their chunk of mod [B, mod_ld]."""
    check(lib().mgx_ln_modulate_fwd(ptr(x.t), x.ld, x.rpb, x.bstride, shift.data_ptr(), scale.data_ptr(), mod_ld, ptr(y),
                                    D, ptr(stats), x.M, D, stream()))


def ln_modulate_bwd(dy, x: Rows, scale, mod_ld, dx: Rows, accumulate, dshift, dscale, D):
    ws = scratch("ln_bwd", lib().mgx_ln_modulate_bwd_workspace(x.M, min(x.rpb, x.M), D), F32, dy.device)
    check(lib().mgx_ln_modulate_bwd(ptr(dy), D, ptr(x.t), x.ld, min(x.rpb, x.M), x.bstride, scale.data_ptr(), mod_ld,
                                    ptr(dx.t), dx.ld, min(dx.rpb, dx.M), dx.bstride, int(accumulate), dshift.data_ptr(),
                                    dscale.data_ptr(), ptr(ws), x.M, D, stream()))


LN2, LOG2E = 0.6931471805599453, 1.4426950408889634
# Q leaves mgx_qk_norm_rope_fwd_qs multiplied by softmax scale * log2(e) (scores = exponents of two: mgx_attn_fwd_log2's
# accumulator-initialised softmax); every other consumer of that Q takes scale = ln 2.  MGX_ATTN_Q_PRESCALE=0: plain Q.
Q_PRESCALE = os.environ.get("MGX_ATTN_Q_PRESCALE", "1") != "0"


def qk_norm_rope(qkv, wq, wk, cos, sin, Q, K, Vt, B, H, S, Sp, rows_per_batch, s0, V=None, Qt=None, Kt=None, q_scale=1.0):
    check(lib().mgx_qk_norm_rope_fwd_qs(ptr(qkv), qkv.shape[-1], ptr(wq), ptr(wk), ptr(cos), ptr(sin), ptr(Q), ptr(K),
                                        ptr(Vt), ptr(V), ptr(Qt), ptr(Kt), B, H, S, Sp, rows_per_batch, s0, float(q_scale),
                                        stream()))


def qk_norm_rope_bwd(qkv, wq, wk, cos, sin, dQ, dK, dV, dqkv, gwq, gwk, B, H, S, Sp, rows_per_batch, s0, ld_dqkv=None,
                     q_scale=1.0):
    """`dqkv`: a tensor whose data_ptr() is the first element of the [rows, >= 3d] output; `ld_dqkv` its row stride in
    elements (default: 3d, a plain matrix).  `q_scale`: the forward's (dQ is the gradient of the scaled Q)."""
    ws = scratch("qk_bwd", lib().mgx_qk_norm_rope_bwd_workspace(B, H, rows_per_batch), F32, qkv.device)
    check(lib().mgx_qk_norm_rope_bwd_qs(ptr(qkv), qkv.shape[-1], ptr(wq), ptr(wk), ptr(cos), ptr(sin), ptr(dQ), ptr(dK),
                                        ptr(dV), dqkv.data_ptr(), qkv.shape[-1] if ld_dqkv is None else ld_dqkv, ptr(gwq),
                                        ptr(gwk), ptr(ws), B, H, S, Sp, rows_per_batch, s0, float(q_scale), stream()))


def linear_qk_norm_rope(X, Wqk, bias, wq, wk, cos, sin, Q, K, B, H, S, rows_per_batch, s0, Kdim, q_scale=1.0, pairs=None):
    """Q, K [B, H, S, 128] <- qk_norm_rope(X @ Wqk^T + bias) in ONE launch (`mgx_linear_qk_norm_rope`: the norm / RoPE / head
    split run in the GEMM's epilogue).  X plain [B * rows_per_batch, Kdim], Wqk [2 * H * 128, Kdim].  False -- nothing launched --
    when the persistent kernel cannot take the problem: the caller keeps `gemm` + `qk_norm_rope`.  `pairs`: [S, 64, 2] (cos, sin)
    per rotation pair when both entries of every pair of `cos` / `sin` are equal (`rope_pair_table`): half the table bytes."""
    prof = GEMM_PROFILE is not None
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = lib().mgx_linear_qk_norm_rope(ptr(X), ptr(Wqk), ptr(bias), ptr(wq), ptr(wk), ptr(cos), ptr(sin), ptr(pairs), ptr(Q), ptr(K),
                                       B, H, S, rows_per_batch, s0, Kdim, Kdim, Kdim, float(q_scale), stream())
    if rc == 1:
        return False
    check(rc)
    if prof:
        e1.record()
        GEMM_PROFILE.append((e0, e1, 2.0 * B * rows_per_batch * 2 * H * 128 * Kdim, (B * rows_per_batch, 2 * H * 128, Kdim, 5)))
    return True


def rope_pair_table(cos, sin):
    """[S, 64, 2] (cos, sin) per rotation pair if the two entries of every pair are equal in both tables, else None."""
    if not (torch.equal(cos[:, 0::2], cos[:, 1::2]) and torch.equal(sin[:, 0::2], sin[:, 1::2])):
        return None
    return torch.stack([cos[:, 0::2], sin[:, 0::2]], dim=-1).contiguous()


def attn_fwd(Q, K, Vt, O_ptr_tensor, lse, B, H, S, Sp, ldo, o_bstride, scale):
    check(lib().mgx_attn_fwd(ptr(Q), ptr(K), ptr(Vt), O_ptr_tensor.data_ptr(), ptr(lse), B, H, S, Sp, ldo, o_bstride,
                             scale, stream()))


def attn_fwd_log2(Q2, K, Vt, O_ptr_tensor, lse, B, H, S, Sp, ldo, o_bstride):
    """Q2 = Q * scale * log2(e), from qk_norm_rope(..., q_scale=scale * LOG2E)."""
    check(lib().mgx_attn_fwd_log2(ptr(Q2), ptr(K), ptr(Vt), O_ptr_tensor.data_ptr(), ptr(lse), B, H, S, Sp, ldo, o_bstride,
                                  stream()))


def attn_fp8_quantize(Q, K, Vt, Q8, K8, V8t, amax, B, H, S, Sp):
    """Per-(batch, head) e4m3 quantisation of the attention operands (csrc/attention_fp8.hip)."""
    check(lib().mgx_attn_fp8_quantize(ptr(Q), ptr(K), ptr(Vt), ptr(Q8), ptr(K8), ptr(V8t), ptr(amax), B, H, S, Sp,
                                      stream()))


def attn_fwd_fp8(Q8, K8, V8t, amax, O_ptr_tensor, lse, B, H, S, Sp, ldo, o_bstride, scale):
    check(lib().mgx_attn_fwd_fp8(ptr(Q8), ptr(K8), ptr(V8t), ptr(amax), O_ptr_tensor.data_ptr(), ptr(lse), B, H, S, Sp,
                                 ldo, o_bstride, scale, stream()))


def skinny_linear(x, W, bias, out, N, K):
    """out[b] = bf16(x[b] @ W^T + bias) for <= 16 rows per call (more rows are chunked)."""
    Bn = x.shape[0]
    for b0 in range(0, Bn, 16):
        nb = min(16, Bn - b0)
        check(lib().mgx_skinny_linear(x[b0:].data_ptr(), x.stride(0), ptr(W), K, ptr(bias), out[b0:].data_ptr(),
                                      out.stride(0), nb, N, K, stream()))


def skinny_wgrad(dout, x, dW, dbias, N, K):
    Bn = x.shape[0]
    for b0 in range(0, Bn, 16):
        nb = min(16, Bn - b0)
        check(lib().mgx_skinny_wgrad(dout[b0:].data_ptr(), dout.stride(0), x[b0:].data_ptr(), x.stride(0), ptr(dW), K,
                                     ptr(dbias), nb, N, K, stream()))


def skinny_dgrad(dout, W, dx, N, K, accumulate=True):
    """dx[Bn, K] (+)= bf16(dout[Bn, N] @ W[N, K]) from the row-major bf16 weight."""
    Bn = dout.shape[0]
    for b0 in range(0, Bn, 8):
        nb = min(8, Bn - b0)
        ws = scratch("skinny_dgrad", lib().mgx_skinny_dgrad_workspace(nb, K), F32, dout.device)
        check(lib().mgx_skinny_dgrad(dout[b0:].data_ptr(), dout.stride(0), ptr(W), K, dx[b0:].data_ptr(), dx.stride(0),
                                     ptr(ws), nb, N, K, 1 if accumulate else 0, stream()))


def ew(a, b, y, op):
    check(lib().mgx_ew_bf16(ptr(a), ptr(b), ptr(y), a.numel(), op, stream()))


def sincos_embed(t, out):
    check(lib().mgx_sincos_embed(ptr(t), ptr(out), t.numel(), stream()))


def cast_bf16(x, y):
    check(lib().mgx_cast_f32_bf16(ptr(x), ptr(y), x.numel(), stream()))


def gelu_rows(x, ldx, y, ldy, M, N):
    """y[m, :N] = bf16(gelu_tanh(x[m, :N])) for M rows `ldx` / `ldy` elements apart: the bias+GELU epilogue's activation
    re-created from a kept pre-activation (bit-identical, see include/mixgrpo_hip.h)."""
    check(lib().mgx_gelu_bf16(ptr(x), ldx, ptr(y), ldy, M, N, stream()))


def cast_f32(x, y, scale=1.0):
    """y (fp32) = scale * x (bf16)."""
    check(lib().mgx_cast_bf16_f32(ptr(x), ptr(y), x.numel(), float(scale), stream()))


def gate_bwd(dout: Rows, y, gate, gate_ld, dy, dgate, batches, rows_per_batch, D):
    ws = scratch("gate_bwd", lib().mgx_gate_bwd_workspace(batches, rows_per_batch, D), F32, y.device)
    check(lib().mgx_gate_bwd(ptr(dout.t), dout.ld, dout.bstride, ptr(y), D, gate.data_ptr(), gate_ld, ptr(dy), D,
                             dgate.data_ptr(), ptr(ws), batches, rows_per_batch, D, stream()))


def sqnorm(g, out, beta=0.0):
    ws = scratch("sqnorm", lib().mgx_sqnorm_workspace(), torch.float64, g.device)
    check(lib().mgx_sqnorm_f32(ptr(g), g.numel(), ptr(ws), ptr(out), beta, stream()))


def adamw_step(w, w16, g, m, v, lr, beta1, beta2, eps, wd, step, gnorm_sq, max_norm, grad_scale=1.0):
    check(lib().mgx_adamw_step(ptr(w), ptr(w16), ptr(g), ptr(m), ptr(v), w.numel(), lr, beta1, beta2, eps, wd, step,
                               ptr(gnorm_sq), max_norm, grad_scale, stream()))


def attn_bwd(Q, K, V, Qt, Kt, O, dO, lse, delta, dOt, dQ, dK, dV, B, H, S, Sp, ldo, o_bstride, scale):
    check(lib().mgx_attn_bwd(ptr(Q), ptr(K), ptr(V), ptr(Qt), ptr(Kt), O.data_ptr(), dO.data_ptr(), ptr(lse), ptr(delta),
                             ptr(dOt), ptr(dQ), ptr(dK), ptr(dV), B, H, S, Sp, ldo, o_bstride, scale, stream()))


# ------------------------------------------------------------------------------------------------ VAE decode (csrc/vae.hip)
def conv3x3(xpad, Wt, bias, out, H, W, C, Cout, ones=None, ld_out=None):
    """out[H W, Cout] (+)= conv3x3(xpad) + bias: xpad zero-bordered NHWC [(H+2), (W+2), C], Wt [Cout, 3, 3, C].
    `ones` given: the residual form out = bf16(out + bf16(conv + bias))."""
    check(lib().mgx_conv3x3_nhwc(ptr(xpad), ptr(Wt), ptr(bias), ptr(out), Cout if ld_out is None else ld_out, ptr(ones), H, W, C,
                                 Cout, 0 if ones is None else 1, stream()))


def group_norm(x, gamma, beta, out, H, W, C, G, silu, padded, eps=1e-6):
    """out = [silu](GroupNorm(x)), x plain [H W, C]; `padded`: out is a zero-bordered [(H+2), (W+2), C] image (interior written)."""
    ws = scratch("group_norm", lib().mgx_group_norm_workspace(H * W, C, G), F32, x.device)
    if padded:
        o, out_row = out.view(-1)[(W + 2) * C + C:], (W + 2) * C
    else:
        o, out_row = out, W * C
    check(lib().mgx_group_norm_nhwc(ptr(x), C, ptr(gamma), ptr(beta), o.data_ptr(), out_row, C, ptr(ws), H, W, C, G, eps,
                                    1 if silu else 0, stream()))


def upsample2x_pad(x, outpad, H, W, C):
    """plain [H W, C] -> interior of the zero-bordered [(2H+2), (2W+2), C] image `outpad`"""
    check(lib().mgx_upsample2x_pad_nhwc(ptr(x), outpad.view(-1)[(2 * W + 2) * C + C:].data_ptr(), H, W, C, stream()))


def latents_to_pad(z, outpad, Cin, H, W, C):
    check(lib().mgx_latents_to_pad_nhwc(ptr(z), outpad.view(-1)[(W + 2) * C + C:].data_ptr(), Cin, H, W, C, stream()))


def nhwc_to_image(x, ld, img, Cout, H, W):
    check(lib().mgx_nhwc_to_image(ptr(x), ld, ptr(img), Cout, H, W, stream()))


def softmax_rows(S, P, M, n, scale):
    check(lib().mgx_softmax_rows_f32(ptr(S), S.stride(0), ptr(P), P.stride(0), M, n, scale, stream()))
