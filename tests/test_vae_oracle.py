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
