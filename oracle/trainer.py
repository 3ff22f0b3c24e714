"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product.

Restates the reference's GRPO train step (group rollout -> group-relative advantages -> PPO-clip
loss -> backward / optimizer), pinned against tests/golden/trainer.* which were produced by running
the reference's own `train_one_step` in this container on CPU+gloo with a toy velocity model.

Follows (relative to /root/reference/fastvideo):
  prepare_latent_image_ids / pack_latents / unpack_latents   train_grpo_flux.py:80-115
  grpo_one_step                                              train_grpo_flux.py:118-181
  sample_reference_model (minus VAE decode / reward models)  train_grpo_flux.py:184-329
  gather_tensor                                              train_grpo_flux.py:332-338
  train_one_step (advantages :439-501, loss :536-615)        train_grpo_flux.py:341-624
  balance_pos_neg                                            models/reward_model/utils.py:18-48
The VAE decode and the reward models are out of scope: `reward_fn(i, latents)`
stands in for decode+score and returns (total: list[float], per_head: dict[str, list[float]]).
"""
import random

import torch
import torch.distributed as dist

from . import rollout as R
from . import solver as S


def prepare_latent_image_ids(height, width, dtype):
    ids = torch.zeros(height, width, 3)
    ids[..., 1] += torch.arange(height)[:, None]
    ids[..., 2] += torch.arange(width)[None, :]
    return ids.reshape(height * width, 3).to(dtype)


def pack_latents(lat):
    b, c, h, w = lat.shape
    return lat.view(b, c, h // 2, 2, w // 2, 2).permute(0, 2, 4, 1, 3, 5).reshape(b, (h // 2) * (w // 2), c * 4)


def unpack_latents(lat, height, width, vae_scale_factor=8):
    b, n, ch = lat.shape
    h = 2 * (int(height) // (vae_scale_factor * 2))
    w = 2 * (int(width) // (vae_scale_factor * 2))
    return lat.view(b, h // 2, w // 2, ch // 4, 2, 2).permute(0, 3, 1, 4, 2, 5).reshape(b, ch // 4, h, w)


def _world():
    return dist.get_world_size() if dist.is_initialized() else 1


def gather_tensor(t):
    if not dist.is_initialized():
        return t
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.cat(out, dim=0)


def _avg(t):
    t = t.detach().clone()
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.AVG)
    return t.item()


def sample_group(args, transformer, ehs, pooled, text_ids, window, reward_fn, injected=None):
    """G sequential batch-1 rollouts sharing x_T when init_same_noise (train_grpo_flux.py:184-329)."""
    T = args.sampling_steps
    sigmas = S.sd3_time_shift(args.shift, torch.linspace(1, 0, T + 1))
    lh, lw = args.h // 8, args.w // 8
    B = ehs.shape[0]
    if args.init_same_noise:
        x_T = injected["x_T"] if injected else torch.randn((1, 16, lh, lw), dtype=torch.bfloat16)
    lat_all, lp_all, ids_all, tot_all, head_all = [], [], [], [], {}
    for i in range(B):
        if not args.init_same_noise:
            x_T = torch.randn((1, 16, lh, lw), dtype=torch.bfloat16)
        z0 = pack_latents(x_T)
        ids = prepare_latent_image_ids(lh // 2, lw // 2, torch.bfloat16)
        if args.training_strategy == "part":
            det = [j not in set(window) for j in range(T)]
        elif args.training_strategy == "all":
            det = [False] * T
        with torch.no_grad():
            _, latents, bl, blp = R.run_sample_step(args, z0, range(T), sigmas, transformer, ehs[i:i + 1],
                                                    pooled[i:i + 1], text_ids[i:i + 1], ids, True, det,
                                                    noises=[n[i:i + 1] for n in injected["steps"]] if injected else None)
        lat_all.append(bl)
        lp_all.append(blp)
        ids_all.append(ids)
        tot, heads = reward_fn(i, latents)
        tot_all.append(torch.tensor(tot, dtype=torch.float32))
        for k, v in heads.items():
            head_all.setdefault(k, []).append(torch.tensor(v, dtype=torch.float32))
    if args.multi_reward_mix == "reward_aggr":
        rewards = torch.cat(tot_all, dim=0)
    else:
        rewards = {k: torch.cat(v, dim=0) for k, v in head_all.items()}
    return rewards, torch.cat(lat_all, 0), torch.cat(lp_all, 0), sigmas, torch.stack(ids_all, 0)


def group_advantages(r, G, trimmed_ratio):
    """Per group of G: (r - mean) / (unbiased std + 1e-8), optionally with trimmed statistics (:447-461)."""
    adv = torch.zeros_like(r)
    for g in range(len(r) // G):
        grp = r[g * G:(g + 1) * G]
        stat = grp
        if trimmed_ratio > 0:
            srt = torch.sort(grp)[0]
            stat = srt[min(int(len(srt) * trimmed_ratio), len(srt) - 1):]
        adv[g * G:(g + 1) * G] = (grp - stat.mean()) / (stat.std() + 1e-8)
    return adv


def balance_pos_neg(samples, use_random=False):
    if use_random:
        return random.sample(samples, len(samples))
    pos = [s for s in samples if s["advantages"].item() > 0]
    neg = [s for s in samples if s["advantages"].item() < 0]
    pos = random.sample(pos, len(pos))
    neg = random.sample(neg, len(neg))
    small, large = (pos, neg) if len(pos) < len(neg) else (neg, pos)
    out = []
    for a, b in zip(small, large):
        out += [a, b]
    return out + large[len(small):]


def replay_log_prob(args, transformer, latents, next_latents, ehs, pooled, text_ids, image_ids, timesteps, index,
                    sigmas):
    """grpo_one_step (train_grpo_flux.py:118-181): model in train mode, log-prob of the stored transition."""
    transformer.train()
    pred = transformer(hidden_states=latents, encoder_hidden_states=ehs, timestep=timesteps / 1000,
                       guidance=torch.tensor([3.5], dtype=torch.bfloat16), txt_ids=text_ids.repeat(ehs.shape[1], 1),
                       pooled_projections=pooled, img_ids=image_ids.squeeze(0), joint_attention_kwargs=None,
                       return_dict=False)[0]
    if args.dpm_algorithm_type == "null" or ("dpmsolver" in args.dpm_algorithm_type
                                             and args.dpm_apply_strategy == "post"):
        if args.flow_grpo_sampling:
            return S.flow_grpo_step(pred, latents.to(torch.float32), args.eta, sigmas, index,
                                    next_latents.to(torch.float32))[2]
        return S.dance_grpo_step(pred, latents.to(torch.float32), args.eta, sigmas, index,
                                 next_latents.to(torch.float32), True, True)[2]
    return S.dpm_step(args, pred, latents.to(torch.float32), index, sigmas[:-1], sigmas, dpm_state=None,
                      generator=torch.Generator(), sde_solver=True)[2]


def train_one_step(args, transformer, optimizer, lr_scheduler, batch, reward_fn, reward_weights, window,
                   max_grad_norm, trace=None, injected=None):
    """Returns (total_loss, grad_norm, policy_total_loss, kl_total_loss, total_clip_frac, reward_mean)."""
    tot = pol = klt = clip_t = 0.0
    optimizer.zero_grad()
    ehs, pooled, text_ids, caption = batch
    G = args.num_generations
    if args.use_group:
        ehs, pooled, text_ids = (torch.repeat_interleave(t, G, dim=0) for t in (ehs, pooled, text_ids))
    rewards, all_lat, all_lp, sigmas, all_ids = sample_group(args, transformer, ehs, pooled, text_ids, window, reward_fn,
                                                             injected=injected)
    B = all_lat.shape[0]
    T = args.sampling_steps
    tsv = [int(s * 1000) for s in sigmas][:T]
    ts = torch.tensor([tsv] * B, dtype=torch.long)
    smp = {"timesteps": ts[:, :-1], "latents": all_lat[:, :-1][:, :-1], "next_latents": all_lat[:, 1:][:, :-1],
           "log_probs": all_lp[:, :-1], "image_ids": all_ids, "text_ids": text_ids,
           "encoder_hidden_states": ehs, "pooled_prompt_embeds": pooled}

    if args.multi_reward_mix == "advantage_aggr":
        gathered = {k: gather_tensor(v.to(torch.float32)) for k, v in rewards.items()}
        if not args.use_group:
            raise ValueError("advantage_aggr needs use_group")
        adv = torch.zeros_like(next(iter(rewards.values())))
        for k, v in rewards.items():
            adv += group_advantages(v.to(torch.float32), G, args.trimmed_ratio) * reward_weights[k]
    elif args.multi_reward_mix == "reward_aggr":
        r = rewards.to(torch.float32)
        gathered = gather_tensor(r)
        adv = group_advantages(r, G, args.trimmed_ratio) if args.use_group else \
            (r - gathered.mean()) / (gathered.std() + 1e-8)
    else:
        raise ValueError(args.multi_reward_mix)
    smp["advantages"] = adv
    if trace is not None:
        trace["advantages"] = adv.clone()
        trace["log_probs"] = all_lp.clone()
        trace["all_latents"] = all_lat.clone()

    perms = None
    if args.training_strategy == "all":
        perms = torch.stack([torch.randperm(len(smp["timesteps"][0])) for _ in range(B)])
        rows = torch.arange(B)[:, None]
        for k in ("timesteps", "latents", "next_latents", "log_probs"):
            smp[k] = smp[k][rows, perms]
    per = [dict(zip(smp, x)) for x in zip(*[v.unsqueeze(1) for v in smp.values()])]
    if args.training_strategy == "part":
        train_ts = list(window)
        if args.advantage_rerange_strategy == "random":
            per = balance_pos_neg(per, use_random=True)
        elif args.advantage_rerange_strategy == "balance":
            per = balance_pos_neg(per, use_random=False)
        elif args.advantage_rerange_strategy != "null":
            raise ValueError(args.advantage_rerange_strategy)
    else:
        n = args.frozen_init_timesteps if args.frozen_init_timesteps > 0 else \
            int(len(smp["timesteps"][0]) * args.timestep_fraction)
        train_ts = list(range(n))

    grad_norm = None
    denom = args.gradient_accumulation_steps * len(train_ts)
    new_lps = []
    for i, s in enumerate(per):
        for t in train_ts:
            new_lp = replay_log_prob(args, transformer, s["latents"][:, t], s["next_latents"][:, t],
                                     s["encoder_hidden_states"], s["pooled_prompt_embeds"], s["text_ids"],
                                     s["image_ids"], s["timesteps"][:, t],
                                     perms[i][t] if perms is not None else t, sigmas)
            new_lps.append(new_lp.detach().clone())
            a = torch.clamp(s["advantages"], -args.adv_clip_max, args.adv_clip_max)
            old = s["log_probs"][:, t]
            ratio = torch.exp(new_lp - old)
            unclipped = -a * ratio
            clipped = -a * torch.clamp(ratio, 1.0 - args.clip_range, 1.0 + args.clip_range)
            clip_frac = torch.mean((torch.abs(ratio - 1.0) > args.clip_range).float())
            policy = torch.mean(torch.maximum(unclipped, clipped)) / denom
            kl = 0.5 * torch.mean((new_lp - old) ** 2) / denom
            loss = policy + args.kl_coeff * kl
            loss.backward()
            tot += _avg(loss)
            pol += _avg(policy)
            klt += _avg(kl)
            clip_t += _avg(clip_frac)
        if (i + 1) % args.gradient_accumulation_steps == 0:
            grad_norm = transformer.clip_grad_norm_(max_grad_norm)
            optimizer.step()
            lr_scheduler.step()
            optimizer.zero_grad()
        if dist.is_initialized():
            dist.barrier()
    if trace is not None:
        trace["new_log_probs"] = new_lps
    rm = {k: v.mean().item() for k, v in gathered.items()} if isinstance(gathered, dict) else gathered.mean().item()
    return tot, (grad_norm.item() if grad_norm is not None else None), pol, klt, clip_t, rm
