"""BASELINE.json configs[0] on the HIP path: "2-layer random-init MMDiT, 64x64 latent, 8-step sampler / 2-step SDE window,
group_size=4, constant dummy reward" -- read as 1 double + 1 single block at FLUX.1-dev's full width (d = 3072, 24 heads,
4096-wide text states, 768-wide pooled vector), 64x64 latent (N = 1024 image tokens) + 512 text tokens.  The product has no
CPU path, so the configuration's own `CPU gloo world_size=1` leg is the oracle's (tests/test_config0_cpu.py); this is the
same plumbing through the HIP engine (reference train_grpo_flux.py:341-624), whose known answer is exact: a constant reward
gives advantages (r - mean) / (0 + 1e-8) = 0 (App. C-12), hence loss 0, gradient 0, nothing clipped, and AdamW moves every
weight by the decoupled weight-decay factor only, once per optimizer step (G / accum = 2 of them)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("stream_k", [False, True])
def test_config0_constant_reward_on_the_hip_engine(stream_k, monkeypatch):
    from mixgrpo_amd import ops
    from mixgrpo_amd import train_grpo_flux as TG
    monkeypatch.setattr(ops, "GEMM_STREAM_K", stream_k)
    from mixgrpo_amd.flux import FluxConfig, FluxTransformer2DModel
    from mixgrpo_amd.grpo_states import GRPOTrainingStates
    from mixgrpo_amd.optim import ConstantWithWarmup, FusedAdamW
    dev = torch.device("cuda", 0)
    cfg = FluxConfig(num_layers=1, num_single_layers=1)                       # everything else: FLUX.1-dev
    assert cfg.dim == 3072 and cfg.num_attention_heads == 24
    m = FluxTransformer2DModel(cfg, device=dev).init_synthetic(seed=0, std=0.02)
    lr, wd = 1e-3, 1e-2
    opt = FusedAdamW(m, lr=lr, betas=(0.9, 0.999), weight_decay=wd, eps=1e-8)
    args = TG.default_args(h=512, w=512, sampling_steps=8, num_generations=4, gradient_accumulation_steps=2)
    states = GRPOTrainingStates(iters_per_group=25, group_size=2, max_timesteps=8 - 2, prog_overlap=True,
                                prog_overlap_step=1, roll_back=True)
    window = states.get_current_timesteps()
    assert list(window) == [0, 1]
    g = torch.Generator().manual_seed(714)
    batch = ((0.1 * torch.randn(1, 512, 4096, generator=g)).bfloat16().to(dev), torch.randn(1, 768, generator=g).bfloat16().to(dev),
             torch.zeros(1, 3, device=dev), ["a prompt"])
    torch.manual_seed(714)
    w_before = m.store.w32.clone()

    def const_reward(latents, captions):
        n = latents.shape[0]
        assert latents.shape == (4, 1024, 64)                                 # G samples of a packed 64x64 latent
        return [0.5] * n, {"Const": [0.5] * n}

    trace = {}
    res = TG.train_one_step(args, dev, m, None, const_reward, opt, ConstantWithWarmup(opt, 0), iter([batch]), None, 1.0,
                            window, 0, {"Const": 1.0}, trace=trace)
    torch.cuda.synchronize()
    assert trace["log_probs"].shape == (4, 8) and torch.isfinite(trace["log_probs"][:, [0, 1]]).all()
    assert torch.equal(trace["advantages"].cpu(), torch.zeros(4))
    assert res[0] == 0.0 and res[2] == 0.0 and res[4] == 0.0                  # loss, policy loss, clip fraction
    assert res[1] == 0.0                                                      # zero-gradient plumbing check
    assert 0.0 <= res[3] < 1e-8                                               # logged KL term (kl_coeff 0): replay == rollout
    assert res[5] == {"Const": 0.5}
    assert len(trace["grad_norms"]) == 2                                      # G / accum optimizer steps
    # first replay chunk: the rollout policy itself (same kernels, same weights).  The reference runs rollout and replay at batch
    # 1; here the rollout runs the group as one batch and the replay a micro-batch, i.e. GEMMs of different M.  With every tile
    # computed whole (no stream-K workspace) each output row is the same K-loop whatever M is: BIT-IDENTICAL log-probs.  With
    # the stream-K tail (the default) the tiles of a launch's last round are summed in parts, and which tiles those are depends
    # on M: the fp32 summation order of some rows differs, a few bf16 roundings flip, and the log-probs agree to ~1e-6 -- two
    # orders of magnitude inside clip_range (1e-4) and three inside the north star's 1e-3.
    pairs, new_lp = trace["new_log_probs"][0]
    old = torch.stack([trace["log_probs"][i, t] for i, t in pairs])
    if not stream_k:
        assert torch.equal(new_lp, old)
    else:
        print(f"\nstream-K: first replay chunk vs rollout log-probs, max |diff| = {(new_lp - old).abs().max().item():.3e}")
        assert (new_lp - old).abs().max().item() < 2e-5
    # AdamW with an all-zero gradient: m = v = 0, update 0 / (0 + eps) = 0 -> w <- w * (1 - lr * wd), twice
    w_after = m.store.w32
    assert torch.allclose(w_after, w_before * (1 - lr * wd) ** 2, rtol=1e-6, atol=1e-12)
    assert not torch.equal(w_after, w_before)
    assert torch.equal(m.store.w16, w_after.to(torch.bfloat16))               # the bf16 compute mirror follows the master
    assert opt.m.abs().max().item() == 0.0 and opt.v.abs().max().item() == 0.0
