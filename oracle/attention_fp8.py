"""CPU restatement of the fp8 (OCP e4m3) attention forward of mixgrpo_amd/csrc/attention_fp8.hip.

TEST INFRASTRUCTURE ONLY (imported by tests/ alone).  PARITY UNPINNED against the reference: zqqqqz2000/MixGRPO has no
fp8 attention (its attention is bf16 F.scaled_dot_product_attention, fastvideo/utils/sampling_utils.py:68-82,
train_grpo_flux.py:134-144); the fp8 path exists because BASELINE.json configs[4] names one.  What is pinned instead:
the quantiser bit-exactly (torch's float8_e4m3fn cast is the OCP round-to-nearest-even conversion the hardware's
v_cvt_pk_fp8_f32 performs), the contractions exactly on power-of-two probabilities, and the whole operator against bf16
attention within the e4m3 tolerance stated in tests/test_hip_attention_fp8.py.
"""
import math

import torch

F8_MAX = 448.0


def amax_table(Q, K, V):
    """[3, B*H] fp32: max |x| of Q, K, V [B, H, S, hd] per (batch, head)."""
    B, H = Q.shape[:2]
    return torch.stack([t.float().abs().reshape(B * H, -1).amax(dim=1) for t in (Q, K, V)])


def quantize(x, amax):
    """x [B, H, S, hd] bf16 -> float8_e4m3fn of x * (448 / amax) (fp32 arithmetic, IEEE division, saturated)."""
    B, H = x.shape[:2]
    sc = (torch.tensor(F8_MAX, dtype=torch.float32) / amax.clamp_min(1e-30)).view(B, H, 1, 1)
    return (x.float() * sc).clamp(-F8_MAX, F8_MAX).to(torch.float8_e4m3fn)


def key_order():
    """Position p of a 64-key block -> key: p = 32h + 16kb + i  <-  32kb + 8(i >> 2) + 4h + (i & 3)."""
    order = []
    for p in range(64):
        h, kb, i = p >> 5, (p >> 4) & 1, p & 15
        order.append(32 * kb + 8 * (i >> 2) + 4 * h + (i & 3))
    return order


def v8t_layout(v8, Sp):
    """v8 [B, H, S, hd] float8 -> uint8 [B, H, hd, Sp]: transposed, zero-padded to Sp keys, keys of every 64-block in
    `key_order()`."""
    B, H, S, hd = v8.shape
    vt = torch.zeros(B, H, hd, Sp, dtype=torch.uint8)
    vt[..., :S] = v8.view(torch.uint8).transpose(2, 3)
    idx = torch.tensor(key_order())
    blocks = vt.view(B, H, hd, Sp // 64, 64)
    return blocks[..., idx].reshape(B, H, hd, Sp).contiguous()


def attention(Q, K, V, scale=None, quantize_p=False):
    """softmax(scale * Q K^T) V with e4m3 operands and exact (fp64) softmax / accumulation.  Returns (O [B,H,S,hd]
    fp64, lse [B,H,S] fp64 natural log).  `quantize_p`: additionally round P / max(P) * 4 to e4m3 (what the kernel
    feeds to the P V contraction when a row's running maximum is final; an approximation of its tile-wise maxima)."""
    hd = Q.shape[-1]
    scale = 1.0 / math.sqrt(hd) if scale is None else scale
    am = amax_table(Q, K, V)
    B, H = Q.shape[:2]
    deq = lambda x8, a: x8.to(torch.float64) * (a.double().clamp_min(1e-30) / F8_MAX).view(B, H, 1, 1)
    q, k, v = deq(quantize(Q, am[0]), am[0]), deq(quantize(K, am[1]), am[1]), deq(quantize(V, am[2]), am[2])
    s = torch.einsum("bhqd,bhkd->bhqk", q, k) * scale
    lse = torch.logsumexp(s, dim=-1)
    p = torch.exp(s - s.amax(dim=-1, keepdim=True))
    den = p.sum(-1, keepdim=True)
    if quantize_p:
        p = (p * 4.0).float().to(torch.float8_e4m3fn).to(torch.float64) / 4.0
    return torch.einsum("bhqk,bhkd->bhqd", p, v) / den, lse
