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


def test_diagnostic_library_is_refused(tmp_path, monkeypatch):
    """A library built with -DMGX_DIAGNOSTIC_BUILD (timing-only switches: wrong results) reports a negative mgx_version();
    `_lib.lib()` must refuse it.  A stand-in with that one symbol is enough: the version is checked before anything binds."""
    import subprocess

    import pytest
    src = tmp_path / "stub.c"
    src.write_text("int mgx_version(void) { return -101; }\n")
    so = tmp_path / "libstub.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)], check=True)
    monkeypatch.setattr(_lib, "LIB_PATH", str(so))
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.MgxError, match="DIAGNOSTIC"):
        _lib.lib()


def test_in_tree_build_takes_no_flags_from_the_environment(monkeypatch):
    """build.py's default target is always the plain sources: diagnostic flags are only accepted together with an output
    path under scratch/."""
    import pytest
    from mixgrpo_amd import build as B
    monkeypatch.setenv("MGX_BUILD_EXTRA", "-DMGX_TIMING_ONLY_NO_EPILOGUE")
    assert not any("TIMING_ONLY" in f for f in B.COMMON)
    with pytest.raises(RuntimeError):
        B.build(diagnostic_flags=["-DMGX_TIMING_ONLY_NO_EPILOGUE"])
    with pytest.raises(RuntimeError):
        B.build(diagnostic_out="/tmp/not_scratch.so", diagnostic_flags=["-DMGX_TIMING_ONLY_NO_EPILOGUE"])
