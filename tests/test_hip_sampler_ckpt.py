"""GPU tests of the rows next to the hot path: the mixed-model inference sampler against the CPU oracle
loop, and the checkpoint / resume files."""
import json
import os

import pytest
import torch

from oracle import mmdit as OM
from oracle import sampler as OS

pytestmark = pytest.mark.gpu

CFG = dict(num_layers=1, num_single_layers=2, attention_head_dim=128, num_attention_heads=4, joint_attention_dim=64,
           pooled_projection_dim=32)


def _pair(seed):
    from mixgrpo_amd.flux import FluxConfig, FluxTransformer2DModel
    P = OM.init_params(OM.FluxConfig(**CFG), seed=seed, std=0.05, bias_std=0.05)
    m = FluxTransformer2DModel(FluxConfig(**CFG), device="cuda")
    m.load_state_dict({k: v.cuda() for k, v in P.items()})
    return P, m.eval()


def test_dual_sampler_vs_oracle(tmp_path):
    from mixgrpo_amd.sample_flux import DualFluxSampler
    P_base, m_base = _pair(1)
    P_new, m_new = _pair(2)
    g = torch.Generator().manual_seed(0)
    B, L, hw = 2, 24, 128                                   # 128 px -> 8 x 8 latent tokens
    ehs = torch.randn(B, L, 64, generator=g).bfloat16()
    pooled = torch.randn(B, 32, generator=g).bfloat16()
    lat = torch.randn(B, 64, 64, generator=g).bfloat16()
    ids = torch.zeros(8, 8, 3)
    ids[..., 1] += torch.arange(8)[:, None]
    ids[..., 2] += torch.arange(8)[None]
    ref = OS.dual_sample(P_base, P_new, OM.FluxConfig(**CFG), lat.clone(), ehs, pooled, torch.zeros(L, 3),
                         ids.reshape(64, 3), num_inference_steps=6, mix_sampling_steps=2)
    s = DualFluxSampler(m_base, m_new)
    out = s(ehs.cuda(), pooled.cuda(), height=hw, width=hw, num_inference_steps=6, mix_sampling_steps=2,
            latents=lat.cuda())
    assert out.dtype == torch.bfloat16 and out.shape == (B, 64, 64)
    err = ((out.float().cpu() - ref.float()).norm() / ref.float().norm()).item()
    assert err < 2e-2, err                                   # six bf16 forwards with fp32 accumulation orders of their own
    # the switch matters: all-base and all-new trajectories differ from the mixed one
    base_only = DualFluxSampler(m_base)(ehs.cuda(), pooled.cuda(), height=hw, width=hw, num_inference_steps=6,
                                        mix_sampling_steps=0, latents=lat.cuda())
    assert ((base_only.float() - out.float()).norm() / out.float().norm()).item() > 5 * err
    with pytest.raises(ValueError):
        DualFluxSampler(m_base)(ehs.cuda(), pooled.cuda(), height=hw, width=hw, mix_sampling_steps=3, latents=lat.cuda())
    # load_new_model reads the tuned weights from a checkpoint file in diffusers key names
    from mixgrpo_amd.checkpoint import save_checkpoint
    d = save_checkpoint(m_new, 0, str(tmp_path), 7, 0)
    s2 = DualFluxSampler(m_base)
    s2.load_new_model(os.path.join(d, "diffusion_pytorch_model.safetensors"))
    out2 = s2(ehs.cuda(), pooled.cuda(), height=hw, width=hw, num_inference_steps=6, mix_sampling_steps=2,
              latents=lat.cuda())
    assert torch.equal(out2, out)


def test_dual_sampler_image_outputs_through_the_hip_vae():
    """The pipeline's tail (sample_flux.py:387-393): unpack, un-scale, `vae.decode`, postprocess -- on the HIP VAE, against the
    oracle VAE applied to the sampler's own latents."""
    from mixgrpo_amd.sample_flux import DualFluxSampler
    from mixgrpo_amd.vae import AutoencoderKL, VaeConfig
    from oracle import vae as OV
    P_base, m_base = _pair(1)
    vkw = dict(block_out_channels=(64, 64, 128, 128), layers_per_block=1, sample_size=256)          # 8x, like the FLUX VAE
    VP = OV.init_params(OV.VaeConfig(**vkw), seed=4)
    vae = AutoencoderKL(VaeConfig(**vkw), device="cuda").load_state_dict({k: v.bfloat16() for k, v in VP.items()})
    g = torch.Generator().manual_seed(0)
    B, L, hw = 2, 24, 128
    ehs = torch.randn(B, L, 64, generator=g).bfloat16()
    pooled = torch.randn(B, 32, generator=g).bfloat16()
    lat = torch.randn(B, 64, 64, generator=g).bfloat16()
    s = DualFluxSampler(m_base, vae=vae)
    kw = dict(height=hw, width=hw, num_inference_steps=3, mix_sampling_steps=0, latents=lat.cuda())
    out_lat = s(ehs.cuda(), pooled.cuda(), **kw)
    img = s(ehs.cuda(), pooled.cuda(), output_type="pt", **kw)
    assert img.shape == (B, 3, hw, hw) and img.dtype == torch.float32 and 0.0 <= img.min().item() and img.max().item() <= 1.0
    # (denormalised in bf16, the decoder's dtype, like VaeImageProcessor.postprocess -- then .float())
    want = (OV.decode_latents(VP, OV.VaeConfig(**vkw), out_lat.float().cpu(), hw, hw).bfloat16() / 2 + 0.5).clamp(0, 1).float()
    assert (img.cpu() - want).abs().mean().item() < 5e-3
    arr = s(ehs.cuda(), pooled.cuda(), output_type="np", **kw)
    assert arr.shape == (B, hw, hw, 3) and arr.dtype.name == "float32"
    pil = s(ehs.cuda(), pooled.cuda(), output_type="pil", **kw)
    assert len(pil) == B and pil[0].size == (hw, hw)
    with pytest.raises(ValueError, match="needs a VAE"):
        DualFluxSampler(m_base)(ehs.cuda(), pooled.cuda(), output_type="pt", **kw)


def test_checkpoint_and_resume_roundtrip(tmp_path):
    from mixgrpo_amd.checkpoint import load_resume_state, save_checkpoint, save_resume_state
    from mixgrpo_amd.flux import FluxTransformer2DModel
    from mixgrpo_amd.grpo_states import GRPOTrainingStates
    from mixgrpo_amd.optim import ConstantWithWarmup, FusedAdamW
    P, m = _pair(3)
    opt = FusedAdamW(m, lr=1e-3)
    sched = ConstantWithWarmup(opt, 4)
    st = GRPOTrainingStates(iters_per_group=2, group_size=2, max_timesteps=6, prog_overlap=True, prog_overlap_step=1,
                            roll_back=True)
    g = m.store.ensure_grad()
    for _ in range(3):                                       # three optimizer / scheduler / window steps
        g.normal_(0, 1e-2, generator=None)
        opt.step(max_grad_norm=1.0)
        sched.step()
        st.update_iteration()
    d = save_checkpoint(m, 0, str(tmp_path), 3, 0)
    assert d.endswith("checkpoint-3-0")
    cfg = json.load(open(os.path.join(d, "config.json")))
    assert cfg["num_layers"] == 1 and "dtype" not in cfg and cfg["_class_name"] == "FluxTransformer2DModel"
    save_resume_state(d, opt, sched, st, global_step=3)
    m2 = FluxTransformer2DModel.from_pretrained(d, device="cuda")
    assert torch.equal(m2.store.w32, m.store.w32) and torch.equal(m2.store.w16, m.store.w16)
    opt2 = FusedAdamW(m2, lr=1e-3)
    sched2 = ConstantWithWarmup(opt2, 0)
    st2 = GRPOTrainingStates(iters_per_group=2, group_size=2, max_timesteps=6)
    assert load_resume_state(d, opt2, sched2, st2) == 3
    assert opt2.step_count == 3 and torch.equal(opt2.m, opt.m) and torch.equal(opt2.v, opt.v)
    assert sched2.get_last_lr() == sched.get_last_lr() and st2 == st
    # the continued run and the resumed run take the same next step, bit for bit
    g2 = m2.store.ensure_grad()
    g.normal_(0, 1e-2)
    g2.copy_(g)
    for o, s_, w in ((opt, sched, st), (opt2, sched2, st2)):
        o.step(max_grad_norm=1.0)
        s_.step()
        w.update_iteration()
    assert torch.equal(m2.store.w32, m.store.w32) and st2.get_current_timesteps() == st.get_current_timesteps()


def test_reward_adapter_decode_contract():
    """reward_adapter: the latents handed to `vae.decode` are unpack(packed) / 0.3611 + 0.1159 in [n, 16, h/8, w/8] layout
    (reference train_grpo_flux.py:284-288), and the per-model / weighted rewards follow the trainer's call-site contract."""
    from mixgrpo_amd.latents import pack_latents
    from mixgrpo_amd.reward_adapter import make_reward_function
    n, h, w = 3, 64, 96
    z = torch.randn(n, 16, h // 8, w // 8, generator=torch.Generator().manual_seed(0)).cuda()
    packed = pack_latents(z, n, 16, h // 8, w // 8)
    seen = {}

    class FakeVAE:
        def enable_tiling(self):
            seen["tiling"] = True

        def decode(self, lat, return_dict=False):
            seen["lat"] = lat.clone()
            return (lat.float().mean(dim=(1, 2, 3)),)          # one "image" scalar per sample

    models = {"HPSClipRewardModel": lambda imgs, prompts: [float(i) for i in imgs],
              "PickScoreRewardModel": lambda imgs, prompts: [float(len(p)) for p in prompts]}
    weights = {"HPSClipRewardModel": 1.0, "PickScoreRewardModel": 0.5}
    fn = make_reward_function(FakeVAE(), models, weights, h, w)
    total, per = fn(packed, ["a", "bb", "ccc"])
    assert seen["tiling"] and seen["lat"].shape == (n, 16, h // 8, w // 8)
    assert torch.allclose(seen["lat"].float(), z / 0.3611 + 0.1159, rtol=1e-6, atol=1e-6)
    assert per["PickScoreRewardModel"] == [1.0, 2.0, 3.0] and len(per["HPSClipRewardModel"]) == n
    for t, a, b in zip(total, per["HPSClipRewardModel"], per["PickScoreRewardModel"]):
        assert abs(t - (a + 0.5 * b)) < 1e-6
