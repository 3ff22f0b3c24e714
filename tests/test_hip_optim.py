"""Direct parity of the fused optimizer kernels (csrc/optim.hip: mgx_sqnorm_f32, mgx_adamw_step) with what the reference
runs per optimizer step -- `transformer.clip_grad_norm_(max_grad_norm)` then `torch.optim.AdamW.step()`
(fastvideo/train_grpo_flux.py:606-607, optimizer built at :715-721: betas (0.9, 0.999), eps 1e-8, weight decay) -- over
several steps on a random ~1 M-element buffer: clip active and inactive, the data-parallel `grad_scale` 1 and 1/8, weight
decay on.  Masters and moments <= 1e-6 relative (fp32 both sides, different operation order), the bf16 compute mirror
exactly bf16(master)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("max_norm,grad_scale,gstd", [(1.0, 1.0, 1e-2),      # |g| ~ 10: clip active (coef ~ 0.1)
                                                       (1.0, 0.125, 1e-5),    # |g/8| << 1: clip inactive, DP scale 1/8
                                                       (0.05, 0.125, 1e-3),   # both: scale then clip
                                                       (None, 1.0, 1e-3)])    # no clipping at all
def test_adamw_and_sqnorm_vs_torch(max_norm, grad_scale, gstd):
    from mixgrpo_amd import ops
    n = 1 << 20
    lr, b1, b2, eps, wd = 2e-4, 0.9, 0.999, 1e-8, 1e-2
    g0 = torch.Generator(device="cuda").manual_seed(5)
    w = torch.randn(n, device="cuda", generator=g0) * 0.02
    ref_p = torch.nn.Parameter(w.clone())
    opt = torch.optim.AdamW([ref_p], lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd)
    w16 = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    m = torch.zeros(n, device="cuda")
    v = torch.zeros(n, device="cuda")
    nsq = torch.zeros(1, device="cuda")
    for step in range(1, 4):                                   # bias corrections at steps 1, 2, 3
        g = torch.randn(n, device="cuda", generator=g0) * gstd * step
        # reference: the summed gradient is averaged over ranks first (grad_scale = 1 / world), then clipped, then stepped
        ref_p.grad = (g * grad_scale).clone()
        total = None
        if max_norm is not None:
            total = torch.nn.utils.clip_grad_norm_([ref_p], max_norm)
        opt.step()
        # product: sum g^2 of the UNSCALED local buffer, scale + clip folded into the AdamW pass
        ops.sqnorm(g, nsq)
        if total is not None:
            assert abs(nsq.sqrt().item() * grad_scale - total.item()) <= 2e-6 * total.item(), (nsq.sqrt().item() * grad_scale, total.item())
        ops.adamw_step(w, w16, g, m, v, lr, b1, b2, eps, wd, step, nsq if max_norm is not None else None,
                       0.0 if max_norm is None else max_norm, grad_scale)
        st = opt.state[ref_p]
        assert _rel(w, ref_p.data) < 1e-6, (step, _rel(w, ref_p.data))
        assert _rel(m, st["exp_avg"]) < 1e-6 and _rel(v, st["exp_avg_sq"]) < 2e-6
        assert (w - ref_p.data).abs().max().item() < 1e-7      # lr * O(1) update, fp32 rounding only
        assert torch.equal(w16, w.to(torch.bfloat16))
    assert not torch.equal(w, torch.randn(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5)) * 0.02)


def test_sqnorm_accumulates_with_beta():
    from mixgrpo_amd import ops
    g = torch.randn(3 * 4096 + 4, device="cuda")
    out = torch.full((1,), 2.0, device="cuda")
    ops.sqnorm(g, out, beta=1.0)
    assert abs(out.item() - (2.0 + g.double().pow(2).sum().item())) < 1e-3
