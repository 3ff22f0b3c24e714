"""BASELINE.json configs[0] on the CPU: "2-layer random-init MMDiT, 64x64 latent, 8-step sampler / 2-step SDE window,
group_size=4, constant dummy reward, CPU gloo world_size=1 (plumbing, no GPU)".  The product has no CPU path, so this is
the ORACLE's train step (oracle/trainer.py driving oracle/mmdit.py, 1 double + 1 single block at a reduced width so the
case runs in seconds) inside a one-rank gloo group: the whole plumbing -- rollout, reward gather, advantages, replay,
loss, clip, optimizer, logging all-reduces -- with a constant reward, whose known answer is exact: all advantages 0, loss 0,
gradient norm 0, nothing clipped, and AdamW moves the weights by the weight-decay term only."""
import os
import tempfile
from argparse import Namespace

import torch
import torch.distributed as dist

from helpers import oracle_flux
from oracle import mmdit as OM
from oracle import trainer as OT


class _Sched:
    def step(self):
        pass


def test_config0_constant_reward_plumbing():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # a file:// store: no TCPStore server outlives this test in the pytest process (the next gloo test binds its own port)
    store_file = tempfile.NamedTemporaryFile(prefix="mgx_cfg0_store_", delete=False)
    store_file.close()
    dist.init_process_group("gloo", init_method=f"file://{store_file.name}", rank=0, world_size=1)
    try:
        a = Namespace(w=512, h=512, t=1, sampling_steps=8, shift=3.0, init_same_noise=True, training_strategy="part",
                      output_dir="/tmp/x", experiment_name="t", reward_model="const", multi_reward_mix="advantage_aggr",
                      use_group=True, num_generations=4, trimmed_ratio=0.0, advantage_rerange_strategy="null",
                      clip_range=1e-4, adv_clip_max=5.0, kl_coeff=0.0, gradient_accumulation_steps=2,
                      frozen_init_timesteps=-1, timestep_fraction=1.0, dpm_algorithm_type="null", dpm_apply_strategy="post",
                      dpm_post_compress_ratio=0.4, dpm_solver_order=2, dpm_solver_type="midpoint",
                      sample_strategy="progressive", flow_grpo_sampling=True, eta=0.7, drop_last_sample=False)
        cfg = OM.FluxConfig(num_layers=1, num_single_layers=1, attention_head_dim=128, num_attention_heads=2,
                            joint_attention_dim=64, pooled_projection_dim=32)
        P = OM.init_params(cfg, seed=0, std=0.02)
        m = oracle_flux(cfg, P)
        lr, wd = 1e-3, 1e-2
        opt = torch.optim.AdamW(m.parameters(), lr=lr, betas=(0.9, 0.999), weight_decay=wd, eps=1e-8)
        before = [p.detach().clone() for p in m.parameters()]
        g = torch.Generator().manual_seed(714)
        batch = ((0.1 * torch.randn(1, 16, 64, generator=g)).bfloat16(), torch.randn(1, 32, generator=g).bfloat16(),
                 torch.zeros(1, 3), ["a prompt"])
        torch.manual_seed(714)
        trace = {}
        res = OT.train_one_step(a, m, opt, _Sched(), batch, lambda i, lat: ([0.5], {"Const": [0.5]}), {"Const": 1.0},
                                [3, 4], 1.0, trace=trace)
        assert trace["all_latents"].shape == (4, 9, 1024, 64)                 # G, T + 1, 64x64 latent packed 2x2, 64 ch
        assert trace["log_probs"].shape == (4, 8) and torch.isfinite(trace["log_probs"][:, [3, 4]]).all()
        assert torch.equal(trace["advantages"], torch.zeros(4))               # constant reward: (r - mean) / (0 + 1e-8) = 0
        assert res[0] == 0.0 and res[2] == 0.0 and res[4] == 0.0
        assert 0.0 <= res[3] < 1e-9                                            # logged KL term (kl_coeff 0): replay == rollout to ~1e-6
        assert res[1] == 0.0                                                   # zero-gradient plumbing check
        assert res[5] == {"Const": 0.5}
        for p, b in zip(m.parameters(), before):                              # two optimizer steps, weight decay only
            assert torch.allclose(p.detach(), b * (1 - lr * wd) ** 2, rtol=1e-6, atol=1e-9)
    finally:
        dist.destroy_process_group()
        os.unlink(store_file.name)
