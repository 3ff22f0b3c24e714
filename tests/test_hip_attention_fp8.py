"""GPU parity of the fp8 (e4m3) attention forward (csrc/attention_fp8.hip, BASELINE.json configs[4]) through the C ABI.

The reference has no fp8 attention, so the checker is oracle/attention_fp8.py (parity unpinned against the reference,
see its header).  What is asserted:
  * the quantiser (amax table, Q8 / K8 bytes, key-permuted V8t bytes) BIT-EXACTLY;
  * the two contractions EXACTLY (up to the bf16 output rounding, 2^-8 relative) on inputs whose scaled scores are
    integers, so every probability is a power of two and e4m3 holds it exactly: this pins the MFMA operand layouts,
    the key permutation, the ragged-tail masking and the LSE;
  * on random inputs the LSE to 2e-3 absolute (the scores carry no P rounding; what remains is the scaled MFMA's
    internal accumulation of the 64 products of an instruction, measured at ~2^-14 of the score magnitude: 6e-4 at
    |score| ~ 15, exact on the small integers of the previous test), and the output within the e4m3
    tolerance: relative L2 <= 4e-2 against the oracle with exact P (P has 3 mantissa bits: rms relative error 3.6 % per
    probability, which random zero-mean V does not average out), and against full-precision attention of the unquantised
    operands no further than the oracle's own distance (the cost of e4m3 Q / K / V) + 4e-2;
  * the MMDiT with attention_dtype="fp8": forward within 6e-2 relative L2 of the bf16-attention forward, and the
    trainer's replayed log-probs BIT-identical to the rollout's on unchanged weights (ratio exactly 1)."""
import math

import pytest
import torch

from oracle import attention_fp8 as OA

pytestmark = pytest.mark.gpu


def _run(Q, K, V, scale=None, want_lse=True):
    """Q, K, V [B, H, S, 128] bf16 (CPU) -> (O [B, H, S, 128] fp32, lse, amax, Q8, K8, V8t) from the HIP path."""
    from mixgrpo_amd import ops
    B, H, S, hd = Q.shape
    Sp = (S + 63) // 64 * 64
    dev = "cuda"
    Qd, Kd = Q.to(dev).contiguous(), K.to(dev).contiguous()
    Vt = torch.zeros(B, H, hd, Sp, dtype=torch.bfloat16, device=dev)
    Vt[..., :S] = V.to(dev).transpose(2, 3)
    u8 = lambda *s: torch.empty(*s, dtype=torch.uint8, device=dev)
    Q8, K8, V8t = u8(B, H, S, hd), u8(B, H, S, hd), u8(B, H, hd, Sp)
    amax = torch.empty(3 * B * H, dtype=torch.float32, device=dev)
    ops.attn_fp8_quantize(Qd, Kd, Vt, Q8, K8, V8t, amax, B, H, S, Sp)
    O = torch.zeros(B, S, H * hd, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=dev) if want_lse else None
    ops.attn_fwd_fp8(Q8, K8, V8t, amax, O, lse, B, H, S, Sp, H * hd, S * H * hd,
                     1.0 / math.sqrt(hd) if scale is None else scale)
    torch.cuda.synchronize()
    Oh = O.float().cpu().view(B, S, H, hd).permute(0, 2, 1, 3)
    return Oh, (lse.cpu() if want_lse else None), amax.cpu().view(3, B * H), Q8.cpu(), K8.cpu(), V8t.cpu()


@pytest.mark.parametrize("B,H,S", [(2, 3, 300), (1, 2, 64), (1, 1, 1), (1, 2, 4608 + 37)])
def test_quantiser_is_bit_exact(B, H, S):
    g = torch.Generator().manual_seed(S)
    Q = (torch.randn(B, H, S, 128, generator=g) * 1.7).bfloat16()
    K = (torch.randn(B, H, S, 128, generator=g) * 0.6).bfloat16()
    V = (torch.randn(B, H, S, 128, generator=g) * torch.rand(B, H, 1, 1, generator=g) * 30).bfloat16()
    Q[0, 0, 0, :4] = torch.tensor([0.0, -0.0, 1e-6, -3e-5]).bfloat16()       # zeros, signed zero, e4m3 subnormals
    _, _, amax, Q8, K8, V8t = _run(Q, K, V, want_lse=False)
    am = OA.amax_table(Q, K, V)
    assert torch.equal(amax, am)
    assert torch.equal(Q8, OA.quantize(Q, am[0]).view(torch.uint8))
    assert torch.equal(K8, OA.quantize(K, am[1]).view(torch.uint8))
    assert torch.equal(V8t, OA.v8t_layout(OA.quantize(V, am[2]), (S + 63) // 64 * 64))


@pytest.mark.parametrize("S", [64, 200, 256 + 64 + 5, 1500])
def test_contractions_exact_on_power_of_two_probabilities(S):
    """scale * log2(e) = 1 and q.k an integer in {0, -1, ..., -6}: P is a power of two, exactly representable in e4m3, so
    O = sum P V8 / sum P holds up to the fp32 accumulation order and the bf16 output rounding."""
    B, H = 1, 2
    g = torch.Generator().manual_seed(S)
    Q = torch.zeros(B, H, S, 128)
    K = torch.zeros(B, H, S, 128)
    Q[..., 0] = torch.randint(1, 3, (B, H, S), generator=g).float()          # 1 or 2
    Q[..., 2] = 3.5                                                           # amax 3.5 -> scale 448 / 3.5 = 128 exactly
    K[..., 0] = -torch.randint(0, 4, (B, H, S), generator=g).float()         # 0 .. -3
    K[:, :, 0, 0] = 0.0                                                       # every row's maximum score is 0, in tile 0
    K[..., 1] = 3.5
    V = (torch.randint(-8, 9, (B, H, S, 128), generator=g).float() / 4)      # exact in e4m3 at scale 128 or 224
    V[..., 0, 0] = 3.5                                                        # amax 3.5
    V = V.clamp(-3.5, 3.5)
    Qb, Kb, Vb = Q.bfloat16(), K.bfloat16(), V.bfloat16()
    O, lse, *_ = _run(Qb, Kb, Vb, scale=math.log(2.0))
    s = torch.einsum("bhqd,bhkd->bhqk", Q.double(), K.double())              # integers: log2 of the probabilities
    p = torch.exp2(s)
    ref = (p @ V.double()) / p.sum(-1, keepdim=True)
    assert torch.allclose(O.double(), ref, rtol=2.0 ** -8, atol=1e-6)
    ref_lse = math.log(2.0) * torch.log2(p.sum(-1))
    assert torch.allclose(lse.double(), ref_lse, rtol=0, atol=2e-5)


@pytest.mark.parametrize("B,H,S,amp", [(2, 3, 333, 1.0), (1, 4, 1024, 2.0), (1, 2, 4608, 1.0)])
def test_random_inputs_within_e4m3_tolerance(B, H, S, amp):
    g = torch.Generator().manual_seed(B * 1000 + S)
    Q = (torch.randn(B, H, S, 128, generator=g) * amp).bfloat16()
    K = (torch.randn(B, H, S, 128, generator=g) * amp).bfloat16()
    V = torch.randn(B, H, S, 128, generator=g).bfloat16()
    O, lse, *_ = _run(Q, K, V)
    ref, ref_lse = OA.attention(Q, K, V)
    assert torch.isfinite(O).all()
    assert (lse.double() - ref_lse).abs().max().item() < 2e-3
    rel = ((O.double() - ref).norm() / ref.norm()).item()
    assert rel < 4e-2, rel
    full = torch.softmax(torch.einsum("bhqd,bhkd->bhqk", Q.double(), K.double()) / math.sqrt(128), -1) @ V.double()
    rel_full = ((O.double() - full).norm() / full.norm()).item()
    quant_only = ((ref - full).norm() / full.norm()).item()      # what e4m3 Q / K / V alone cost (4-8 %, more on peaked rows)
    assert rel_full < quant_only + 4e-2, (rel_full, quant_only)


def test_peaked_rows_exercise_the_rescale_paths():
    """Keys whose scores grow along the sequence force a running-maximum raise (and O / l rescale) in every tile; a few
    huge late keys take the deferred-rescale branch's > 2^6 jump."""
    B, H, S = 1, 2, 700
    g = torch.Generator().manual_seed(9)
    Q = torch.randn(B, H, S, 128, generator=g)
    K = torch.randn(B, H, S, 128, generator=g) * 0.2 + Q.mean(dim=2, keepdim=True) * torch.linspace(0, 3, S).view(1, 1, S, 1)
    K[:, :, 650] = Q[:, :, 5] * 4
    V = torch.randn(B, H, S, 128, generator=g)
    Qb, Kb, Vb = Q.bfloat16(), K.bfloat16(), V.bfloat16()
    O, lse, *_ = _run(Qb, Kb, Vb)
    ref, ref_lse = OA.attention(Qb, Kb, Vb)
    assert torch.isfinite(O).all() and torch.isfinite(lse).all()
    assert (lse.double() - ref_lse).abs().max().item() < 5e-3
    assert ((O.double() - ref).norm() / ref.norm()).item() < 4e-2


CFG = dict(num_layers=1, num_single_layers=1, attention_head_dim=128, num_attention_heads=4, joint_attention_dim=64,
           pooled_projection_dim=32)


def test_mmdit_with_fp8_attention_and_bitwise_replay():
    from mixgrpo_amd import train_grpo_flux as TG
    from mixgrpo_amd.flux import FluxConfig, FluxTransformer2DModel
    from mixgrpo_amd.optim import ConstantWithWarmup, FusedAdamW
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    m16 = FluxTransformer2DModel(FluxConfig(**CFG), device=dev).init_synthetic(seed=5, std=0.05, bias_std=0.02)
    m8 = FluxTransformer2DModel(FluxConfig(**CFG), device=dev, attention_dtype="fp8").init_synthetic(seed=5, std=0.05,
                                                                                                    bias_std=0.02)
    B, N, L = 2, 48 * 4, 16
    xs = torch.randn(B, N, 64, generator=g).to(dev)
    ehs = torch.randn(B, L, 64, generator=g).bfloat16().to(dev)
    pooled = torch.randn(B, 32, generator=g).bfloat16().to(dev)
    ids = torch.zeros(12, 16, 3)
    ids[..., 1] += torch.arange(12)[:, None]
    ids[..., 2] += torch.arange(16)[None]
    ids = ids.reshape(N, 3).to(dev)
    t = torch.tensor([0.954, 0.5]).to(dev)
    gd = torch.tensor([3.5]).bfloat16().to(dev)
    txt = torch.zeros(L, 3, device=dev)
    m16.eval(), m8.eval()
    o16 = m16(xs, ehs, t, gd, txt, pooled, ids)[0].float()
    o8 = m8(xs, ehs, t, gd, txt, pooled, ids)[0].float()
    rel = ((o8 - o16).norm() / o16.norm()).item()
    assert 0 < rel < 6e-2, rel                                                # different kernels, same function

    opt = FusedAdamW(m8, lr=0.0)                                              # weights unchanged -> ratio must be exactly 1
    args = TG.default_args(h=48, w=64, sampling_steps=6, num_generations=4, gradient_accumulation_steps=2)
    loader = iter([(ehs[:1], pooled[:1], torch.zeros(1, 3, device=dev), ["p"])])

    def reward(lat, cap):
        r = torch.tensor([0.1, 0.4, 0.2, 0.9])
        return r, {"Synthetic": r}

    trace = {}
    res = TG.train_one_step(args, dev, m8, None, reward, opt, ConstantWithWarmup(opt, 0), loader, None, 1.0, [1, 2], 0,
                            {"Synthetic": 1.0}, trace=trace)
    lp = trace["log_probs"]
    for pairs, new in trace["new_log_probs"]:
        old = torch.stack([lp[i, tt] for i, tt in pairs])
        assert torch.equal(new, old)
    assert res[4] == 0.0 and trace["grad_norms"][0].item() > 0                # nothing clipped, gradients flow (bf16 backward)
