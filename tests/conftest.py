import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # keep libmixgrpo_hip.so in step with the sources (no-op when the object hashes are current)
    try:
        from mixgrpo_amd.build import build
        build(verbose=False)
    except Exception as e:  # noqa: BLE001 - a stale/missing library fails loudly in the tests that need it
        print(f"[conftest] could not (re)build libmixgrpo_hip.so: {e}")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
