"""Embedding-cache layout (mixgrpo_amd/latent_flux_rl_datasets.py) against the format the reference's preprocessing
script writes (fastvideo/data_preprocess/preprocess_flux_embedding.py:86-115): files are created here exactly as that
script does (`torch.save` of `encode_prompt` outputs with a leading batch dimension of 1, `prompt.json` keys)."""
import json
import os

import pytest
import torch
from torch.utils.data import DataLoader, DistributedSampler

from mixgrpo_amd.latent_flux_rl_datasets import LatentDataset, latent_collate_function


def _write(root, n):
    for d in ("prompt_embed", "pooled_prompt_embeds", "text_ids"):
        os.makedirs(os.path.join(root, d))
    items = []
    for i in range(n):
        torch.save(torch.full((1, 512, 4096), float(i), dtype=torch.bfloat16), os.path.join(root, "prompt_embed", f"{i}.pt"))
        torch.save(torch.full((1, 768), float(i), dtype=torch.bfloat16), os.path.join(root, "pooled_prompt_embeds", f"{i}.pt"))
        torch.save(torch.zeros(512, 3, dtype=torch.bfloat16), os.path.join(root, "text_ids", f"{i}.pt"))
        items.append({"prompt_embed_path": f"{i}.pt", "text_ids": f"{i}.pt", "pooled_prompt_embeds_path": f"{i}.pt",
                      "caption": f"caption {i}", "qa": []})
    with open(os.path.join(root, "prompt.json"), "w") as f:
        json.dump(items, f, indent=4)
    return os.path.join(root, "prompt.json")


def test_layout_and_collate(tmp_path):
    path = _write(str(tmp_path), 5)
    ds = LatentDataset(path, num_latent_t=1, cfg_rate=0.0)
    assert len(ds) == 5 and ds.lengths == [1] * 5
    pe, pooled, tid, cap = ds[3]
    assert pe.shape == (512, 4096) and pooled.shape == (768,) and tid.shape == (3,) and cap == "caption 3"
    # the reference's loader construction (train_grpo_flux.py:736-749): DistributedSampler partition, batch 1
    for rank in range(2):
        sampler = DistributedSampler(ds, rank=rank, num_replicas=2, shuffle=True, seed=1223627)
        dl = DataLoader(ds, sampler=sampler, batch_size=1, collate_fn=latent_collate_function, drop_last=True)
        ehs, pooled, text_ids, caption = next(iter(dl))
        assert ehs.shape == (1, 512, 4096) and pooled.shape == (1, 768) and text_ids.shape == (1, 3)
        assert caption[0] == f"caption {int(ehs[0, 0, 0])}"
        assert text_ids[0].repeat(512, 1).shape == (512, 3)          # sampling_utils.py:77
    with pytest.raises(RuntimeError):
        LatentDataset(path, 1, cfg_rate=1.0)[0]
