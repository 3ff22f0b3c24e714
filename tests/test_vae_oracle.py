"""CPU checks of the VAE-decode oracle (oracle/vae.py; parity unpinned against diffusers, see its header): the tiling arithmetic
and the in-place blend semantics of the reference lineage (fastvideo/models/hunyuan/vae/autoencoder_kl_causal_3d.py:384-399,
472-525), the parameter inventory, and that tiled and whole decodes agree away from the seams."""
import torch

from oracle import vae as OV

SMALL = OV.VaeConfig(block_out_channels=(64, 64), layers_per_block=1, sample_size=32)      # tile: 16 latent / 32 px


def test_tile_sizes_of_the_flux_vae():
    ts, tl, ov = OV.tile_sizes(OV.VaeConfig())
    assert (ts, tl, ov) == (1024, 128, 0.25)            # a 1024^2 image (128^2 latent) is exactly one tile: never tiled
    assert OV.tile_sizes(SMALL)[:2] == (32, 16)


def test_parameter_inventory_matches_the_diffusers_layout():
    shp = OV.param_shapes(OV.VaeConfig())
    assert shp["decoder.conv_in.weight"] == (512, 16, 3, 3)
    assert shp["decoder.mid_block.attentions.0.to_q.weight"] == (512, 512)
    assert shp["decoder.up_blocks.2.resnets.0.conv_shortcut.weight"] == (256, 512, 1, 1)
    assert shp["decoder.up_blocks.3.resnets.0.conv_shortcut.weight"] == (128, 256, 1, 1)
    assert "decoder.up_blocks.3.upsamplers.0.conv.weight" not in shp and "decoder.up_blocks.2.upsamplers.0.conv.weight" in shp
    assert shp["decoder.conv_out.weight"] == (3, 128, 3, 3)
    n = sum(torch.Size(s).numel() for s in shp.values())
    assert 49_000_000 < n < 50_000_000                   # the FLUX VAE decoder: 49.5 M parameters


def test_blend_is_in_place_and_linear():
    a = torch.ones(1, 1, 8, 4, dtype=torch.bfloat16)
    b = torch.zeros(1, 1, 8, 4, dtype=torch.bfloat16)
    out = OV.blend_v(a, b, 4)
    assert out is b
    assert b[0, 0, :, 0].tolist() == [1.0, 0.75, 0.5, 0.25, 0.0, 0.0, 0.0, 0.0]
    c = torch.zeros(1, 1, 4, 8, dtype=torch.bfloat16)
    OV.blend_h(torch.ones(1, 1, 4, 8, dtype=torch.bfloat16), c, 100)        # extent clipped to the tile
    assert c[0, 0, 0].tolist() == [1 - x / 8 for x in range(8)]


def test_tiled_decode_layout_and_agreement_away_from_seams():
    P = OV.init_params(SMALL, seed=3)
    g = torch.Generator().manual_seed(4)
    z = torch.randn(1, 16, 20, 28, generator=g)          # 2 x 3 tiles of 16 with stride 12
    whole = OV.decoder(P, SMALL, z)
    calls = []
    tiled = OV.tiled_decode(P, SMALL, z, decode_tile=lambda t: (calls.append(tuple(t.shape[-2:])), OV.decoder(P, SMALL, t))[1])
    assert calls == [(16, 16), (16, 16), (16, 4), (8, 16), (8, 16), (8, 4)]
    assert tiled.shape == (1, 3, 40, 56) and tiled.dtype == torch.bfloat16      # two resolutions: 2x upsampling
    assert OV.decode(P, SMALL, z).shape == tiled.shape and OV.decode(P, SMALL, z[..., :16, :16]).shape == (1, 3, 32, 32)
    # the first tile's top-left corner sees the same receptive field in both decodes (GroupNorm statistics differ per tile,
    # so only roughly)
    d = (tiled[..., :8, :8].float() - whole[..., :8, :8]).abs().mean() / whole[..., :8, :8].abs().mean()
    assert d < 0.5


def test_decode_latents_unpacks_like_the_trainer():
    P = OV.init_params(SMALL, seed=1)
    lat = torch.randn(2, 4, 64, generator=torch.Generator().manual_seed(0))      # 32 x 32 px: 2 x 2 packed tokens
    img = OV.decode_latents(P, SMALL, lat, 32, 32)
    assert img.shape == (2, 3, 8, 8)                      # 4 x 4 latent, this config upsamples 2x (FLUX: 8x)


# ---------------------------------------------------------------------------------------------------------------------------
# Pins: what the reference HOLDS of the VAE decode as plain Python, run in place by tests/golden/gen_fixtures.py `vae`
# (its own `__init__` tile-size rule, `blend_v` / `blend_h`, `spatial_tiled_decode` with a recording decoder stand-in;
# fastvideo/models/hunyuan/vae/autoencoder_kl_causal_3d.py:132-139, 384-399, 472-525).  The oracle AND the product's tiling
# (mixgrpo_amd/vae.py: `_blend`, `AutoencoderKL.tiled_decode`, the constructor's tile sizes -- host code, runs on CPU tensors)
# are held to them bit for bit.  The decoder's convolutions / GroupNorm / attention stay unpinned (diffusers, absent).
import json
import os

from safetensors.torch import load_file

_G = os.path.join(os.path.dirname(__file__), "golden")
VT = load_file(os.path.join(_G, "vae_tiling.safetensors"))
VM = json.load(open(os.path.join(_G, "vae_tiling.json")))


def _standin(tile, up):
    """The fixtures' decoder stand-in (gen_fixtures.vae_tile_standin), restated."""
    x = tile[:, :3].float().repeat_interleave(up, dim=-2).repeat_interleave(up, dim=-1)
    rows = torch.arange(x.shape[-2], dtype=torch.float32).view(*([1] * (x.dim() - 2)), -1, 1)
    return (x + rows / 8).to(torch.bfloat16)


def test_tile_size_rule_vs_the_reference_constructor():
    from mixgrpo_amd.vae import AutoencoderKL, VaeConfig
    for row in VM["tile_sizes"]:
        want = (row["tile_sample_min_size"], row["tile_latent_min_size"], row["tile_overlap_factor"])
        assert OV.tile_sizes(OV.VaeConfig(block_out_channels=(64,) * row["n_blocks"], sample_size=row["sample_size"])) == want
        m = AutoencoderKL(VaeConfig(block_out_channels=(64,) * row["n_blocks"], sample_size=row["sample_size"]), device="cpu")
        assert (m.tile_sample_min_size, m.tile_latent_min_size, m.tile_overlap_factor) == want


def test_blends_vs_the_reference_bit_for_bit():
    from mixgrpo_amd.vae import _blend
    assert len(VM["blend"]) == 8
    for row in VM["blend"]:
        a, b, want = (VT[f"{row['key']}/{n}"] for n in ("a", "b", "out"))
        kind = row["key"][-1]
        got = (OV.blend_v if kind == "v" else OV.blend_h)(a.clone(), b.clone(), row["extent"])
        assert got.dtype == want.dtype and torch.equal(got, want), row
        if want.dtype == torch.bfloat16:                       # the product's tiles are bf16 decoder outputs
            b2 = b.clone()
            got2 = _blend(a.clone(), b2, row["extent"], -2 if kind == "v" else -1)
            assert got2 is b2 and torch.equal(got2, want), row


def test_tiled_decode_vs_the_reference_schedule_and_pixels():
    from mixgrpo_amd.vae import AutoencoderKL, VaeConfig
    for row in VM["tiled"]:
        z5, want5 = VT[row["key"] + "/z"], VT[row["key"] + "/out"]
        z, want = z5[:, :, 0], want5[:, :, 0]                  # the reference's tensors are [B, C, T = 1, H, W]
        tl, ts = row["tile_latent_min_size"], row["tile_sample_min_size"]
        up = ts // tl
        nblk = {2: 2, 4: 3, 8: 4}[up]
        calls = []
        cfg = OV.VaeConfig(latent_channels=4, block_out_channels=(64,) * nblk, sample_size=ts)
        assert OV.tile_sizes(cfg)[:2] == (ts, tl)
        got = OV.tiled_decode(None, cfg, z, decode_tile=lambda t: (calls.append(list(t.shape[-2:])), _standin(t, up))[1])
        assert calls == row["decoder_calls"]
        assert list(got.shape) == [want.shape[0], 3, want.shape[-2], want.shape[-1]] and torch.equal(got, want), row["key"]
        # the product's tiling on the same stand-in (host code; `_decode_batch` is the only device call of `tiled_decode`)
        m = AutoencoderKL(VaeConfig(latent_channels=4, block_out_channels=(64,) * nblk, sample_size=ts), device="cpu")
        calls2 = []
        m._decode_batch = lambda t: (calls2.append(list(t.shape[-2:])), _standin(t, up))[1]
        got2 = m.tiled_decode(z)
        assert calls2 == row["decoder_calls"] and torch.equal(got2, want), row["key"]
