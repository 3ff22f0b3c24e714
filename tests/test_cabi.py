"""The C-ABI library loads (no GPU needed) and exports every symbol include/mixgrpo_hip.h declares."""
import os
import re

from mixgrpo_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            txt = open(os.path.join(ROOT, "include", fn)).read()
            txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
            names |= set(re.findall(r"\b(mgx_[a-z0-9_]+)\s*\(", txt))
    return names


def test_library_exports_all_declared_symbols():
    h = _lib.lib()
    decl = declared_symbols()
    assert len(decl) >= 10
    for name in sorted(decl):
        assert hasattr(h, name), f"{name} declared in include/ but not exported"
    assert decl == set(_lib.SIGNATURES), decl ^ set(_lib.SIGNATURES)
    assert h.mgx_version() >= 100
    assert h.mgx_logp_workspace_elems(2, 4096 * 64) > 0


def test_product_refuses_cpu_tensors():
    import pytest
    import torch
    from mixgrpo_amd import sampling_utils as SU
    sig = SU.sd3_time_shift(3.0, torch.linspace(1, 0, 9))
    with pytest.raises(_lib.MgxError):
        SU.flow_grpo_step(torch.zeros(1, 8, 64, dtype=torch.bfloat16), torch.zeros(1, 8, 64), 0.7, sig, 1, None,
                          noise=torch.zeros(1, 8, 64, dtype=torch.bfloat16))


def test_host_coefficients_match_survey_anchors():
    """std*sqrt(-dt) anchors from SURVEY.md 8c (eta .7, shift 3): T=25 i=0 -> .7, i=1 -> .7145020, i=12 -> .2186189."""
    import torch
    from mixgrpo_amd import sampling_utils as SU
    sig = SU.sd3_time_shift(3.0, torch.linspace(1, 0, 26))
    for i, exp in ((0, 0.7), (1, 0.7145020), (2, 0.5087253), (3, 0.4181935), (12, 0.2186189), (23, 0.1106521)):
        assert abs(SU.flow_coeffs(sig, i, 0.7).sd_value - exp) < 2e-7
    assert [int(s * 1000) for s in sig][:6] == [1000, 986, 971, 956, 940, 923]
