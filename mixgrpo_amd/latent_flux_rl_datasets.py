"""Cached text-embedding dataset of the GRPO trainer: the on-disk layout written by the reference's
fastvideo/data_preprocess/preprocess_flux_embedding.py:86-115 and read by
fastvideo/dataset/latent_flux_rl_datasets.py:19-94.

    <dir>/prompt.json                      list of {"prompt_embed_path", "pooled_prompt_embeds_path", "text_ids", "caption", ...}
    <dir>/prompt_embed/<i>.pt              T5 embeddings   [512, 4096]
    <dir>/pooled_prompt_embeds/<i>.pt      CLIP pooled     [768]
    <dir>/text_ids/<i>.pt                  zeros           [3]  (one row; the sampler repeats it, sampling_utils.py:77)

Same class / collate names and return order as the reference, so `DataLoader(LatentDataset(...),
collate_fn=latent_collate_function)` feeds `train_one_step` unchanged.  Files are opened with
`torch.load(weights_only=True)` only (nothing from a file is executed).  A leading batch dimension of 1 (what
`encode_prompt` returns and the preprocessing script saves) is squeezed so that the collated batch is
[B, 512, 4096] / [B, 768] / [B, 3] as the trainer expects (train_grpo_flux.py:362-367).
"""
import json
import os
import random

import torch
from torch.utils.data import Dataset


# prompt.json key -> sub-directory of the cache that holds the file it names (the on-disk schema, see the module docstring)
_CACHE_FIELDS = (("prompt_embed_path", "prompt_embed"), ("pooled_prompt_embeds_path", "pooled_prompt_embeds"),
                 ("text_ids", "text_ids"))


class LatentDataset(Dataset):
    """`LatentDataset(json_path, num_latent_t, cfg_rate)[i] -> (prompt_embed [512, 4096], pooled [768], text_ids [3],
    caption)`.  `num_latent_t` is accepted for signature compatibility (video datasets use it; images have one frame)."""

    def __init__(self, json_path, num_latent_t, cfg_rate):
        root = os.path.dirname(json_path)
        with open(json_path, "r") as f:
            self.entries = json.load(f)               # file order is sample order (the sampler indexes into it)
        self.field_dirs = {key: os.path.join(root, sub) for key, sub in _CACHE_FIELDS}
        self.cfg_rate = float(cfg_rate)
        self.num_latent_t = num_latent_t
        self.lengths = [e.get("length", 1) for e in self.entries]     # what a length-grouped sampler would read

    def _tensor(self, entry, key):
        # weights_only=True: nothing from a cache file is executed
        return torch.load(os.path.join(self.field_dirs[key], entry[key]), map_location="cpu", weights_only=True)

    def __getitem__(self, idx):
        entry = self.entries[idx]
        if random.random() < self.cfg_rate:
            # the reference substitutes a zero prompt embedding here and then fails on the undefined pooled / id
            # tensors (latent_flux_rl_datasets.py:56-78): prompt dropout is not usable in its GRPO trainer (the
            # shipped scripts pass cfg 0.0), so it is rejected here instead of returning a half-initialised sample
            raise RuntimeError("cfg_rate > 0 (prompt dropout) is not supported by the GRPO data path")
        prompt_embed, pooled, text_ids = (self._tensor(entry, key) for key, _ in _CACHE_FIELDS)
        if prompt_embed.dim() == 3 and prompt_embed.shape[0] == 1:
            prompt_embed = prompt_embed[0]
        if pooled.dim() == 2 and pooled.shape[0] == 1:
            pooled = pooled[0]
        if text_ids.dim() == 2:                    # [512, 3] rows of zeros -> the single [3] row the trainer repeats
            text_ids = text_ids[0]
        return prompt_embed, pooled, text_ids, entry["caption"]

    def __len__(self):
        return len(self.entries)


def latent_collate_function(batch):
    prompt_embeds, pooled_prompt_embeds, text_ids, caption = zip(*batch)
    return (torch.stack(prompt_embeds, dim=0), torch.stack(pooled_prompt_embeds, dim=0), torch.stack(text_ids, dim=0),
            caption)
