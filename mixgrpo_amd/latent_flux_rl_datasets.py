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


class LatentDataset(Dataset):
    def __init__(self, json_path, num_latent_t, cfg_rate):
        self.json_path = json_path
        self.cfg_rate = cfg_rate
        self.datase_dir_path = os.path.dirname(json_path)
        self.prompt_embed_dir = os.path.join(self.datase_dir_path, "prompt_embed")
        self.pooled_prompt_embeds_dir = os.path.join(self.datase_dir_path, "pooled_prompt_embeds")
        self.text_ids_dir = os.path.join(self.datase_dir_path, "text_ids")
        with open(self.json_path, "r") as f:
            self.data_anno = json.load(f)
        self.num_latent_t = num_latent_t
        self.lengths = [item["length"] if "length" in item else 1 for item in self.data_anno]

    @staticmethod
    def _load(path):
        return torch.load(path, map_location="cpu", weights_only=True)

    def __getitem__(self, idx):
        item = self.data_anno[idx]
        if random.random() < self.cfg_rate:
            # the reference substitutes a zero prompt embedding here and then fails on the undefined pooled / id
            # tensors (latent_flux_rl_datasets.py:56-78): prompt dropout is not usable in its GRPO trainer (the
            # shipped scripts pass cfg 0.0), so it is rejected here instead of returning a half-initialised sample
            raise RuntimeError("cfg_rate > 0 (prompt dropout) is not supported by the GRPO data path")
        prompt_embed = self._load(os.path.join(self.prompt_embed_dir, item["prompt_embed_path"]))
        pooled = self._load(os.path.join(self.pooled_prompt_embeds_dir, item["pooled_prompt_embeds_path"]))
        text_ids = self._load(os.path.join(self.text_ids_dir, item["text_ids"]))
        if prompt_embed.dim() == 3 and prompt_embed.shape[0] == 1:
            prompt_embed = prompt_embed[0]
        if pooled.dim() == 2 and pooled.shape[0] == 1:
            pooled = pooled[0]
        if text_ids.dim() == 2:                    # [512, 3] rows of zeros -> the single [3] row the trainer repeats
            text_ids = text_ids[0]
        return prompt_embed, pooled, text_ids, item["caption"]

    def __len__(self):
        return len(self.data_anno)


def latent_collate_function(batch):
    prompt_embeds, pooled_prompt_embeds, text_ids, caption = zip(*batch)
    return (torch.stack(prompt_embeds, dim=0), torch.stack(pooled_prompt_embeds, dim=0), torch.stack(text_ids, dim=0),
            caption)
