"""Pins the non-GEMM arithmetic of oracle/mmdit.py against code that IS in /root/reference.

The FLUX MMDiT is diffusers' (absent: `oracle/mmdit.py` is PARITY UNPINNED as a whole), but the reference carries sibling
models built from the same formulas (witness table).  tests/golden/gen_fixtures.py (`mmdit`) imports the
torch-only witnesses
    fastvideo/models/hunyuan/modules/posemb_layers.py:140-193 (rotation), :267-314 (RoPE tables)
    fastvideo/models/hunyuan/modules/embed_layers.py:99-123                  (sinusoidal embedding)
    fastvideo/models/hunyuan/modules/norm_layers.py:5-60, mochi_hf/norm.py:42-65 (RMSNorm)
    fastvideo/models/hunyuan/modules/modulate_layers.py                       (modulate / gate)
and stores their outputs at FLUX shapes (axes 16/56/56, head_dim 128, 256-wide sinusoid, d = 3072); here the oracle's
`rope_tables / apply_rope / sincos256 / rms_norm / modulate / gated residual` are held to them.  Documented dtype
differences are asserted as such, not hidden.  Still unpinned after this file: the block wiring (which tensors feed which
Linear, text-first concatenation, chunk order of the AdaLN outputs), the attention softmax, GELU-tanh, the bf16 rounding
points of autocast, and the `timestep.to(bf16) * 1000` quantisation."""
import math

import pytest
import torch

from helpers import assert_same, host_matches_fixture_host, load_golden
from oracle import mmdit as OM

T_, M_ = load_golden("mmdit_witness")
EXACT = host_matches_fixture_host()


def eq(a, b, **kw):
    assert_same(a, b, exact=EXACT, **kw)


def _flux_ids():
    """512 text rows of zeros + the 64 x 64 image grid (0, row, col): train_grpo_flux.py:80-91, sampling_utils.py:77."""
    img = torch.zeros(64, 64, 3)
    img[..., 1] += torch.arange(64)[:, None]
    img[..., 2] += torch.arange(64)[None, :]
    return torch.cat([torch.zeros(512, 3), img.reshape(-1, 3)], dim=0)


def test_rope_tables_at_flux_positions():
    ids = _flux_ids()
    cos, sin = OM.rope_tables(ids, (16, 56, 56))
    assert cos.shape == (4608, 128) and cos.dtype == torch.float32
    wc = torch.cat([T_[f"rope/axis{a}/cos"][ids[:, a].long()] for a in range(3)], dim=1)
    ws = torch.cat([T_[f"rope/axis{a}/sin"][ids[:, a].long()] for a in range(3)], dim=1)
    # same formula (1 / theta^(arange(0, dim, 2) / dim), outer(pos, freqs), cos / sin, repeat_interleave(2), concatenation over
    # the axes).  Documented difference: the witness forms the frequencies in float32, FluxPosEmbed / the oracle in float64
    # and round once at the end -> agreement to float32 rounding of an angle of up to 63 rad, not bit for bit.
    assert (cos - wc).abs().max().item() < 2e-5 and (sin - ws).abs().max().item() < 2e-5
    assert torch.equal(cos[:512], torch.ones(512, 128)) and torch.equal(sin[:512], torch.zeros(512, 128))   # text: identity
    # layout: each frequency occupies two adjacent columns (interleaved pairs), axis blocks 16 | 56 | 56
    assert torch.equal(cos[:, 0::2], cos[:, 1::2]) and torch.equal(sin[:, 0::2], sin[:, 1::2])
    # and with the frequencies formed in float32 like the witness, the tables agree bit for bit
    pos = ids.float()
    c32 = []
    for a, dim in enumerate((16, 56, 56)):
        fr = 1.0 / (10000.0 ** (torch.arange(0, dim, 2)[:dim // 2].float() / dim))
        c32.append(torch.outer(pos[:, a], fr).cos().repeat_interleave(2, dim=1))
    eq(torch.cat(c32, dim=1), wc)


@pytest.mark.parametrize("tag", ["f32", "bf16"])
def test_rotation(tag):
    cos, sin = T_["rot/cos"], T_["rot/sin"]
    for which in ("q", "k"):
        x = T_[f"rot/{tag}/x{which}"]                               # [B, S, H, D] as the witness takes it
        out = OM.apply_rope(x.float().transpose(1, 2), cos, sin).transpose(1, 2)     # oracle: [B, H, S, D], fp32 arithmetic
        ref = T_[f"rot/{tag}/o{which}"]
        if tag == "bf16":                                           # the witness casts back to the input dtype; the oracle
            out = out.to(torch.bfloat16)                            # rounds once at SDPA entry: same single rounding
        eq(out.contiguous(), ref)
    assert not torch.equal(T_["rot/f32/oq"], T_["rot/f32/xq"])      # (the rotation is not the identity on this input)


def test_sinusoid():
    out = OM.sincos256(T_["sincos/t"])
    assert out.shape == (9, 256)
    eq(out, T_["sincos/out"])
    # [cos | sin] order (flip_sin_to_cos=True) and frequency 0 is exp(0) = 1: column 0 = cos(t), column 128 = sin(t)
    assert torch.allclose(out[:, 0], torch.cos(T_["sincos/t"])) and torch.allclose(out[:, 128], torch.sin(T_["sincos/t"]))
    assert out[5].tolist() == [1.0] * 128 + [0.0] * 128             # t = 0


def test_rms_norm():
    w = T_["rms/w"]
    x = T_["rms/f32/x"]
    out = OM.rms_norm(x, w)
    eq(out, T_["rms/f32/mochi"])                                    # fp32 in: both witnesses are the same arithmetic
    eq(out, T_["rms/f32/hunyuan"])
    xb = T_["rms/bf16/x"]
    ob = OM.rms_norm(xb.float(), w)                                 # fp32 result (diffusers RMSNorm with an fp32 weight)
    eq(ob.to(torch.bfloat16), T_["rms/bf16/mochi"])                 # MochiRMSNorm = the same, cast to the input dtype at the end
    # documented difference: hunyuan's RMSNorm rounds to bf16 BEFORE the weight multiply (norm_layers.py:56-59)
    h = T_["rms/bf16/hunyuan"]
    assert h.dtype == torch.float32 and (h - ob).abs().max().item() < 0.05 and not torch.equal(h, ob)


def test_modulate_and_gate():
    x, shift, scale = T_["mod/x"], T_["mod/shift"], T_["mod/scale"]
    out = OM.modulate(x, shift.float(), scale.float())              # LN (fp32) * bf16(1 + scale) + shift
    assert T_["mod/out"].dtype == torch.float32
    eq(out, T_["mod/out"])
    y, gate, res = T_["gate/y"], T_["gate/gate"], T_["gate/res"]
    g = OM._bf(res.float() + OM._bf(gate.float()[:, None] * y.float()))        # oracle forward()'s gated_residual
    eq(g.to(torch.bfloat16), T_["gate/out"])
