"""Shared test helpers: fixture loading."""
import json
import os

from safetensors.torch import load_file

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    tensors = load_file(os.path.join(GOLDEN, name + ".safetensors"))
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        meta = json.load(f)
    return tensors, meta
