"""HIP VAE decode (mixgrpo_amd/vae.py, csrc/vae.hip, the implicit-convolution mode of csrc/gemm.hip) against the CPU oracle
(oracle/vae.py: parity unpinned against diffusers, which is not available offline -- see its header for the witnesses inside the
reference).  Tolerances: a convolution / GroupNorm result is a bf16 tensor computed from bf16 operands with fp32 accumulation, so
single ops agree to one bf16 ulp of the result's scale; the whole decoder compounds ~30 such layers."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("H,W,C,Cout", [(16, 16, 64, 64),        # 128x128-tile kernel, one K-tile per tap
                                         (24, 20, 128, 64),       # ragged rows: W not a multiple of anything
                                         (128, 128, 256, 256),    # 256x256 persistent kernel (256 tiles), 4 K-tiles per tap
                                         (64, 128, 512, 512),     # 8 K-tiles per tap
                                         (32, 32, 128, 4)])       # conv_out's shape: 4 output channels, generic epilogue
def test_conv3x3_implicit_gemm_vs_torch(H, W, C, Cout):
    from mixgrpo_amd import ops
    g = torch.Generator().manual_seed(H + C)
    x = torch.randn(C, H, W, generator=g).to(BF16)
    w = (torch.randn(Cout, C, 3, 3, generator=g) / (9 * C) ** 0.5).to(BF16)
    b = (0.1 * torch.randn(Cout, generator=g)).to(BF16)
    want = F.conv2d(x.float()[None], w.float(), b.float(), padding=1)[0]                 # [Cout, H, W] fp32
    xp = torch.zeros(H + 2, W + 2, C, dtype=BF16, device="cuda")
    xp[1:-1, 1:-1] = x.permute(1, 2, 0).cuda()
    wt = w.permute(0, 2, 3, 1).contiguous().cuda()
    bias = torch.zeros(Cout + 8, dtype=BF16, device="cuda")
    bias[:Cout] = b.cuda()
    ld = max(Cout, 8)
    out = torch.zeros(H * W, ld, dtype=BF16, device="cuda")
    ops.conv3x3(xp, wt, bias, out, H, W, C, Cout, ld_out=ld)
    got = out[:, :Cout].float().cpu().view(H, W, Cout).permute(2, 0, 1)
    assert (got - want).abs().max().item() <= 2.0 ** -8 * want.abs().max().item() + 1e-6
    assert (got == want.to(BF16).float()).float().mean().item() > 0.97                   # bit-identical almost everywhere
    if Cout >= 8:                                                                         # residual form: out += conv
        res = torch.randn(H * W, Cout, generator=g).to(BF16)
        out2 = res.clone().cuda()
        ones = torch.ones(Cout, dtype=BF16, device="cuda")
        ops.conv3x3(xp, wt, bias, out2, H, W, C, Cout, ones=ones)
        y = want.to(BF16).float().permute(1, 2, 0).reshape(H * W, Cout)
        want2 = (res.float() + y).to(BF16).float()
        err = (out2.float().cpu() - want2).abs()
        # one bf16 ulp of the conv result (where the two fp32 sums round differently) plus one of the sum
        tol = 2.0 ** -6 * (res.float().abs() + y.abs()).clamp_min(1.0)             # ulp(y) <= 2^-7 |y|, then the sum rounds
        assert (err <= tol).all(), ((err / tol).max().item(), (err > tol).float().mean().item())
        assert (out2.float().cpu() == want2).float().mean().item() > 0.95


@pytest.mark.parametrize("H,W,C,silu,padded", [(16, 16, 64, True, True), (33, 17, 128, True, True), (128, 128, 512, False, False),
                                                (256, 256, 128, True, True)])
def test_group_norm_silu_vs_torch(H, W, C, silu, padded):
    from mixgrpo_amd import ops
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(H * W, C, generator=g) * 2 + 0.5).to(BF16)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(BF16)
    beta = (0.1 * torch.randn(C, generator=g)).to(BF16)
    xn = x.float().t().reshape(1, C, H, W)
    want = F.group_norm(xn, 32, gamma.float(), beta.float(), eps=1e-6)
    if silu:
        want = F.silu(want)
    want = want[0].permute(1, 2, 0)                                                       # [H, W, C] fp32
    if padded:
        out = torch.zeros(H + 2, W + 2, C, dtype=BF16, device="cuda")
    else:
        out = torch.zeros(H * W, C, dtype=BF16, device="cuda")
    ops.group_norm(x.cuda(), gamma.cuda(), beta.cuda(), out, H, W, C, 32, silu, padded)
    got = (out[1:-1, 1:-1] if padded else out.view(H, W, C)).float().cpu()
    assert (got - want).abs().max().item() <= 2.0 ** -7 * want.abs().max().item()
    assert _rel(got, want) < 3e-3
    if padded:                                                                            # the border stays zero
        assert out[0].abs().sum().item() == 0 and out[-1].abs().sum().item() == 0
        assert out[:, 0].abs().sum().item() == 0 and out[:, -1].abs().sum().item() == 0


def test_softmax_rows_and_layout_kernels():
    from mixgrpo_amd import ops
    g = torch.Generator().manual_seed(2)
    S = (torch.randn(300, 1000, generator=g) * 20).cuda()
    P = torch.empty(300, 1000, dtype=BF16, device="cuda")
    ops.softmax_rows(S, P, 300, 1000, 0.25)
    want = torch.softmax(S.float() * 0.25, -1)
    assert (P.float() - want).abs().max().item() <= 2.0 ** -8
    x = torch.randn(6 * 10, 64, generator=g).to(BF16).cuda()
    up = torch.zeros(14, 22, 64, dtype=BF16, device="cuda")
    ops.upsample2x_pad(x, up, 6, 10, 64)
    want = F.interpolate(x.float().view(6, 10, 64).permute(2, 0, 1)[None], scale_factor=2.0, mode="nearest")[0].permute(1, 2, 0)
    assert torch.equal(up[1:-1, 1:-1].float(), want) and up[0].abs().sum().item() == 0 and up[:, -1].abs().sum().item() == 0
    z = torch.randn(16, 5, 7, generator=g).cuda()
    zp = torch.zeros(7, 9, 64, dtype=BF16, device="cuda")
    ops.latents_to_pad(z, zp, 16, 5, 7, 64)
    assert torch.equal(zp[1:-1, 1:-1, :16], z.permute(1, 2, 0).to(BF16)) and zp[..., 16:].abs().sum().item() == 0
    y = torch.randn(5 * 7, 8, generator=g).to(BF16).cuda()
    img = torch.empty(3, 5, 7, dtype=BF16, device="cuda")
    ops.nhwc_to_image(y, 8, img, 3, 5, 7)
    assert torch.equal(img, y[:, :3].t().reshape(3, 5, 7))


def _pair(cfg_kw, seed):
    from oracle import vae as OV
    from mixgrpo_amd.vae import AutoencoderKL, VaeConfig
    ocfg = OV.VaeConfig(**cfg_kw)
    P = OV.init_params(ocfg, seed=seed)
    m = AutoencoderKL(VaeConfig(**cfg_kw), device="cuda").load_state_dict({k: v.to(BF16) for k, v in P.items()})
    return OV, ocfg, P, m


def test_small_decoder_vs_oracle_whole_and_tiled():
    kw = dict(block_out_channels=(64, 128, 128), layers_per_block=1, sample_size=64)       # 4x upsampling; tile 16 latent / 64 px
    OV, ocfg, P, m = _pair(kw, seed=3)
    g = torch.Generator().manual_seed(9)
    z = torch.randn(2, 16, 16, 16, generator=g)
    want = OV.decode(P, ocfg, z)
    m.enable_tiling()
    got = m.decode(z.cuda(), return_dict=False)[0]
    assert got.shape == (2, 3, 64, 64) and got.dtype == BF16
    assert _rel(got.float().cpu(), want.float()) < 2e-2
    z2 = torch.randn(1, 16, 20, 28, generator=g)                                            # 2 x 3 tiles, blended
    want2 = OV.decode(P, ocfg, z2)
    got2 = m.decode(z2.cuda(), return_dict=False)[0]
    assert got2.shape == want2.shape == (1, 3, 80, 112)
    assert _rel(got2.float().cpu(), want2.float()) < 2e-2
    m.disable_tiling()
    whole = m.decode(z2.cuda(), return_dict=False)[0]
    assert _rel(whole.float().cpu(), OV.decoder(P, ocfg, z2)) < 2e-2 and not torch.equal(whole, got2)


def test_blend_matches_the_row_by_row_loop_bit_for_bit():
    from oracle import vae as OV
    from mixgrpo_amd.vae import _blend
    g = torch.Generator().manual_seed(1)
    a = torch.randn(1, 3, 64, 48, generator=g).to(BF16)
    b = torch.randn(1, 3, 64, 48, generator=g).to(BF16)
    for dim, fn in ((-2, OV.blend_v), (-1, OV.blend_h)):
        want = fn(a.clone(), b.clone(), 16)
        got = _blend(a.clone().cuda(), b.clone().cuda(), 16, dim)
        assert torch.equal(got.cpu(), want)


def test_flux_vae_decode_1024_vs_oracle():
    """The headline shape: one 128 x 128 latent (a 1024^2 image) through the FLUX VAE configuration -- exactly one tile, so
    `enable_tiling()` changes nothing (reference lineage: only when a side exceeds the tile)."""
    import json
    import os
    import time
    OV, ocfg, P, m = _pair({}, seed=5)
    g = torch.Generator().manual_seed(6)
    z = torch.randn(1, 16, 128, 128, generator=g)
    want = OV.decode(P, ocfg, z)
    m.enable_tiling()
    got = m.decode(z.cuda(), return_dict=False)[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        m.decode(z.cuda(), return_dict=False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    rel = _rel(got.float().cpu(), want.float())
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump({"decode_1024_ms": ms, "rel_l2_vs_oracle": rel, "max_abs": (got.float().cpu() - want.float()).abs().max().item(),
               "image_abs_mean": want.float().abs().mean().item()}, open("gpurun_out/r03_vae_decode.json", "w"), indent=1)
    assert got.shape == (1, 3, 1024, 1024)
    assert rel < 3e-2, rel


def test_reward_adapter_runs_on_the_hip_vae():
    """The decode + reward stage the trainer calls (mixgrpo_amd/reward_adapter.py = train_grpo_flux.py:279-316) with the HIP VAE
    as its `vae`: packed rollout latents in, per-head rewards out, the decoded images equal to the oracle's `decode_latents`."""
    from oracle import vae as OV
    from mixgrpo_amd.reward_adapter import make_reward_function, decode_latents
    kw = dict(block_out_channels=(64, 64, 128, 128), layers_per_block=1, sample_size=64)     # 8x upsampling like FLUX's VAE
    OV_, ocfg, P, m = _pair(kw, seed=8)
    g = torch.Generator().manual_seed(3)
    h = w = 64
    lat = torch.randn(3, (h // 16) * (w // 16), 64, generator=g)
    want = OV.decode_latents(P, ocfg, lat, h, w)
    imgs = decode_latents(m, lat.cuda(), h, w)
    assert imgs.shape == (3, 3, 64, 64)
    assert _rel(imgs.float().cpu(), want.float()) < 2e-2
    seen = {}

    def brightness(images, prompts):
        seen["n"] = (len(images), tuple(images[0].shape), list(prompts))
        return [float(im.float().mean()) for im in images]

    fn = make_reward_function(m, {"Brightness": brightness}, {"Brightness": 2.0}, h, w)
    total, heads = fn(lat.cuda(), ["a", "b", "c"])
    assert seen["n"] == (3, (3, 64, 64), ["a", "b", "c"])
    assert len(total) == 3 and total[0] == pytest.approx(2.0 * heads["Brightness"][0])
    assert heads["Brightness"][1] == pytest.approx(want[1].float().mean().item(), abs=2e-2)


def test_from_pretrained_reads_the_diffusers_layout(tmp_path):
    """`AutoencoderKL.from_pretrained(path, subfolder="vae", torch_dtype=torch.bfloat16)` as the reference calls it
    (train_grpo_flux.py:697-701): config.json + diffusion_pytorch_model.safetensors with diffusers key names; encoder / quant-conv
    tensors and unknown config entries are ignored, a missing decoder tensor is an error."""
    import json
    import os
    from safetensors.torch import save_file
    from oracle import vae as OV
    from mixgrpo_amd.vae import AutoencoderKL
    kw = dict(block_out_channels=(64, 128), layers_per_block=1, sample_size=32)
    OV_, ocfg, P, m = _pair(kw, seed=2)
    d = tmp_path / "model" / "vae"
    os.makedirs(d)
    json.dump({"_class_name": "AutoencoderKL", "block_out_channels": [64, 128], "layers_per_block": 1, "sample_size": 32,
               "latent_channels": 16, "out_channels": 3, "norm_num_groups": 32, "scaling_factor": 0.3611, "shift_factor": 0.1159,
               "use_quant_conv": False, "force_upcast": True, "act_fn": "silu", "mid_block_add_attention": True},
              open(d / "config.json", "w"))
    sd = {k: v.to(BF16).contiguous() for k, v in P.items()}
    sd["encoder.conv_in.weight"] = torch.zeros(64, 3, 3, 3, dtype=BF16)
    save_file(sd, str(d / "diffusion_pytorch_model.safetensors"))
    m2 = AutoencoderKL.from_pretrained(str(tmp_path / "model"), subfolder="vae", torch_dtype=BF16)
    assert m2.config.block_out_channels == (64, 128) and m2.tile_latent_min_size == 16
    z = torch.randn(1, 16, 12, 12, generator=torch.Generator().manual_seed(1)).cuda()
    assert torch.equal(m2.decode(z, return_dict=False)[0], m.decode(z, return_dict=False)[0])
    assert m2.decode(z)["sample"].shape == (1, 3, 24, 24)
    del sd["decoder.conv_out.bias"]
    save_file(sd, str(d / "diffusion_pytorch_model.safetensors"))
    with pytest.raises(KeyError, match="decoder tensors"):
        AutoencoderKL.from_pretrained(str(tmp_path / "model"), subfolder="vae")
