"""The skinny linears of the temb / AdaLN-modulation path (csrc/small.hip): forward, weight gradient and the input gradient
`mgx_skinny_dgrad`, against autograd of `torch.nn.functional.linear` on the same bf16 operands (what `loss.backward()`,
fastvideo/train_grpo_flux.py:600, runs through diffusers' AdaLayerNormZero.linear under bf16 autocast: fp32 accumulation, a
bf16 result per linear, bf16 accumulation of the per-user gradients of silu(temb))."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


@pytest.mark.parametrize("Bn,N,K", [(7, 6 * 3072, 3072),      # double-block modulation at micro-batch 7
                                     (8, 3 * 3072, 3072),      # single-block modulation
                                     (1, 2 * 3072, 3072),      # norm_out, one sample
                                     (12, 3072, 3072),         # temb MLP, 12 rows: two calls of the 8-row kernel
                                     (3, 130, 264)])           # ragged: rows not a multiple of the 128 row groups, K % 2048 != 0
def test_skinny_dgrad_vs_autograd(Bn, N, K):
    from mixgrpo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(11)
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(BF16)
    dout = torch.randn(Bn, N, device="cuda", generator=g).to(BF16)
    acc0 = torch.randn(Bn, K, device="cuda", generator=g).to(BF16)
    ref = (dout.double() @ W.double())                                  # exact contraction
    want = (acc0.float() + ref.float().to(BF16).float()).to(BF16)      # bf16 result, added to the running bf16 sum
    got = acc0.clone()
    ops.skinny_dgrad(dout, W, got, N, K)
    # fp32 accumulation in a different order than the exact sum: at most one bf16 ulp on the linear's result, then one more
    # on the running sum
    err = (got.float() - want.float()).abs()
    tol = 2.0 ** -7 * want.float().abs().clamp_min(ref.float().abs()) + 1e-6
    assert (err <= tol).all(), (err / tol).max().item()
    assert (got == want).float().mean().item() > 0.98                  # almost all elements bit-identical
    fresh = torch.full((Bn, K), 7.0, device="cuda", dtype=BF16)
    ops.skinny_dgrad(dout, W, fresh, N, K, accumulate=False)           # overwrite form
    assert (fresh.float() - ref.float()).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item() + 1e-6


def test_skinny_linear_and_wgrad_vs_autograd():
    from mixgrpo_amd import ops
    Bn, N, K = 7, 3 * 3072, 3072
    g = torch.Generator(device="cuda").manual_seed(12)
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(BF16)
    b = torch.randn(N, device="cuda", generator=g).to(BF16)
    x = torch.randn(Bn, K, device="cuda", generator=g).to(BF16)
    out = torch.empty(Bn, N, device="cuda", dtype=BF16)
    ops.skinny_linear(x, W, b, out, N, K)
    want = (x.double() @ W.double().t() + b.double())
    assert (out.double() - want).abs().max().item() <= 2.0 ** -8 * want.abs().max().item() + 1e-6
    dout = torch.randn(Bn, N, device="cuda", generator=g).to(BF16)
    dW = torch.randn(N, K, device="cuda", generator=g) * 0.1
    db = torch.randn(N, device="cuda", generator=g) * 0.1
    dW0, db0 = dW.clone(), db.clone()
    ops.skinny_wgrad(dout, x, dW, db, N, K)
    assert torch.allclose(dW, dW0 + (dout.float().t() @ x.float()), rtol=1e-5, atol=1e-5)
    assert torch.allclose(db, db0 + dout.float().sum(0), rtol=1e-5, atol=1e-5)
