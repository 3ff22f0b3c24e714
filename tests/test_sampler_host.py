"""Host-side schedule of the mixed-model inference sampler (mixgrpo_amd/sample_flux.py; reference
fastvideo/sample/sample_flux.py:249-264) against the oracle restatement and hand-computed anchors.  The scheduler
itself lives in diffusers (absent offline): parity unpinned, see oracle/sampler.py."""
import math

import torch

from mixgrpo_amd import sample_flux as SF
from oracle import sampler as OS


def test_calculate_shift_anchors():
    assert abs(SF.calculate_shift(256) - 0.5) < 1e-12           # base_image_seq_len -> base_shift
    assert abs(SF.calculate_shift(4096) - 1.15) < 1e-12         # max_image_seq_len (1024^2) -> max_shift
    assert abs(SF.calculate_shift(2025) - OS.calculate_shift(2025)) < 1e-15   # 720^2


def test_sigma_schedule_matches_oracle_and_formula():
    for T, n in ((28, 4096), (50, 2025), (6, 64)):
        sig, ts = SF.flow_match_sigmas(T, SF.calculate_shift(n))
        ref = OS.sigmas_for(T, n)
        assert torch.equal(sig, ref)
        assert sig.shape == (T + 1,) and sig[0] == 1.0 and sig[-1] == 0.0
        assert torch.all(sig[:-1] > sig[1:])                    # strictly decreasing
        assert torch.equal(ts, sig[:-1] * 1000.0)
        mu = SF.calculate_shift(n)
        s_last = 1.0 / T
        assert abs(sig[T - 1].item() - math.exp(mu) / (math.exp(mu) + (1 / s_last - 1))) < 1e-6


def test_schedule_helpers_pinned_to_the_reference():
    """`calculate_shift` and the sigma grid / mu that reach the scheduler, against values produced by the reference's own
    vendored helpers (fastvideo/models/flux_hf/pipeline_flux.py:73-84,87-145 through tests/golden/gen_fixtures.py sampler;
    call site fastvideo/sample/sample_flux.py:248-264).  Bit for bit in float64."""
    import json
    import os
    import numpy as np
    with open(os.path.join(os.path.dirname(__file__), "golden", "sampler_schedule.json")) as f:
        fx = json.load(f)
    for n, vals in fx["calculate_shift"].items():
        for impl in (SF.calculate_shift, OS.calculate_shift):
            assert impl(int(n)) == vals["default"] == vals["flux_config"], (n, impl(int(n)), vals)
            assert impl(int(n), 256, 4096, 0.5, 1.15) == vals["flux_config"]
    for case in fx["retrieve_timesteps"]:
        T, n_img = case["T"], case["n_img"]
        assert SF.calculate_shift(n_img) == case["mu"] == case["passed"]["mu"]
        assert case["passed"]["num_inference_steps"] is None and case["returned_steps"] == T == case["returned_len"]
        grid = np.asarray(case["passed"]["sigmas"])
        assert np.array_equal(OS.sigma_grid(T).numpy(), grid)                   # the oracle's grid
        sig, ts = SF.flow_match_sigmas(T, case["mu"], sigmas=grid)              # the product takes the same grid ...
        sig2, _ = SF.flow_match_sigmas(T, case["mu"])                           # ... and builds the same one itself
        assert torch.equal(sig, sig2) and sig.shape == (T + 1,) and sig[-1] == 0 and torch.equal(ts, sig[:-1] * 1000.0)


def test_rope_pair_table_only_for_tables_that_repeat_each_pair():
    """`ops.rope_pair_table` (host logic behind mgx_linear_qk_norm_rope's optional (cos, sin)-per-pair table): built from the
    model's own rope tables (`flux.rope_tables`, FluxPosEmbed's repeat_interleave layout: both entries of a rotation pair
    equal), refused for tables where they differ."""
    import torch
    from mixgrpo_amd import ops
    from mixgrpo_amd.flux import rope_tables
    ids = torch.zeros(40, 3)
    ids[:, 1] = torch.arange(40) // 8
    ids[:, 2] = torch.arange(40) % 8
    cos, sin = rope_tables(ids, (16, 56, 56))
    assert cos.shape == (40, 128) and cos.dtype == torch.float32
    pairs = ops.rope_pair_table(cos, sin)
    assert pairs is not None and pairs.shape == (40, 64, 2) and pairs.is_contiguous()
    assert torch.equal(pairs[:, :, 0], cos[:, 0::2]) and torch.equal(pairs[:, :, 1], sin[:, 1::2])
    sin2 = sin.clone()
    sin2[3, 5] += 1e-3
    assert ops.rope_pair_table(cos, sin2) is None
    cos2 = cos.clone()
    cos2[0, 0] -= 1e-3
    assert ops.rope_pair_table(cos2, sin) is None
