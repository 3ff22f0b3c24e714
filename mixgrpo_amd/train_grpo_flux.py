"""MI355X-native GRPO rollout-and-update engine: drop-in for the hot-path functions of the reference trainer
`fastvideo/train_grpo_flux.py` (:80-624): `prepare_latent_image_ids`, `pack_latents`, `unpack_latents`,
`grpo_one_step`, `sample_reference_model`, `gather_tensor`, `train_one_step` -- same names, argument meaning,
return values and error behaviour.

What is different (by design, results unchanged per sample):
  * the G rollouts of a group run as ONE batch through the HIP MMDiT instead of G sequential batch-1 passes;
  * the replayed (sample, window-step) pairs between two optimizer steps run as one (micro)batched
    forward/backward; gradients accumulate in the flat fp32 buffer exactly like the reference's accumulation;
  * sigma tables / coefficients live on the host, rewards / advantages / losses stay on the device: the only
    host syncs are the ones the caller asks for (returned python floats at the end of the step);
  * data parallelism is replica-DP over RCCL (one prompt group per rank per step, like the reference's
    DistributedSampler partitioning): all-gather of rewards, bucketed all-reduce of the flat gradient buffer.
Out of scope here: VAE decode and reward models -- `reward_function(latents, captions)`
stands in for decode+score and returns (total rewards [n], {head: rewards [n]}) as tensors or lists.
"""
import random
from argparse import Namespace

import numpy as np
import torch
import torch.distributed as dist

from . import ops
from . import sampling_utils as SU
from ._lib import check, lib, ptr, stream
from .dist_utils import GradReducer, allreduce_mean_vec_, gather_tensor, is_dist, main_print, rank, world_size
from .latents import pack_latents, prepare_latent_image_ids, unpack_latents  # noqa: F401  (re-exported surface)

F32, BF16 = torch.float32, torch.bfloat16


def default_args(**kw):
    """The flags the hot path reads (reference train_grpo_flux.py:894-1423), script values."""
    a = dict(h=1024, w=1024, t=1, sampling_steps=25, shift=3.0, eta=0.7, init_same_noise=True, use_group=True,
             num_generations=8, training_strategy="part", flow_grpo_sampling=True, drop_last_sample=False,
             clip_range=1e-4, adv_clip_max=5.0, kl_coeff=0.0, gradient_accumulation_steps=3, max_grad_norm=1.0,
             multi_reward_mix="advantage_aggr", trimmed_ratio=0.0, advantage_rerange_strategy="null",
             timestep_fraction=0.6, frozen_init_timesteps=-1, dpm_algorithm_type="null", dpm_apply_strategy="post",
             dpm_post_compress_ratio=0.4, dpm_solver_order=2, dpm_solver_type="midpoint",
             sample_strategy="progressive", output_dir="data/outputs", experiment_name="exp", reward_model="synthetic",
             rollout_batch=0, train_microbatch=0, skip_dead_backward=False)
    a.update(kw)
    return Namespace(**a)


def balance_pos_neg(samples, use_random=False):
    """reference fastvideo/models/reward_model/utils.py:18-48 (python `random`, host logic)."""
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


def _timestep_tensor(values, device):
    """int(sigma*1000)/1000 per row as an IEEE fp32 division done on the host (see sampling_utils.run_sample_step)."""
    v = np.asarray(values, dtype=np.float32) / np.float32(1000)
    return torch.from_numpy(v).to(device)


def grpo_one_step(args, latents, pre_latents, encoder_hidden_states, pooled_prompt_embeds, text_ids, image_ids,
                  transformer, timesteps, i, sigma_schedule):
    """Replay one stored transition with the current policy -> log_prob [B] (differentiable w.r.t. the model
    output); reference train_grpo_flux.py:118-181."""
    dev = latents.device
    transformer.train()
    ts = timesteps
    if ts.dtype in (torch.long, torch.int32, torch.int64):
        ts = _timestep_tensor(ts.detach().cpu().numpy(), dev)
    with torch.autocast("cuda", torch.bfloat16):
        pred = transformer(hidden_states=latents, encoder_hidden_states=encoder_hidden_states, timestep=ts,
                           guidance=torch.tensor([3.5], device=dev, dtype=BF16),
                           txt_ids=text_ids.repeat(encoder_hidden_states.shape[1], 1),
                           pooled_projections=pooled_prompt_embeds,
                           img_ids=image_ids.squeeze(0) if image_ids.dim() == 3 else image_ids,
                           joint_attention_kwargs=None, return_dict=False)[0]
    if args.dpm_algorithm_type == "null" or ("dpmsolver" in args.dpm_algorithm_type
                                             and args.dpm_apply_strategy == "post"):
        if args.flow_grpo_sampling:
            return SU.flow_grpo_step(pred, latents.to(F32), args.eta, sigma_schedule, i, pre_latents.to(F32),
                                     determistic=False, want_x0=False, want_mean=False)[2]
        return SU.dance_grpo_step(pred, latents.to(F32), args.eta, sigma_schedule, i, pre_latents.to(F32), True, True)[2]
    # strategy "all" under DPM (train_grpo_flux.py:170-180): a first-order SDE step from a freshly (default-)seeded
    # generator; the log-prob scores that step's own sample (detached) and differentiates through its mean
    return SU.dpm_step(args, pred, latents.to(F32), i, sigma_schedule[:-1], sigma_schedule, dpm_state=None,
                       generator=torch.Generator(device=dev), sde_solver=True)[2]


def _as_tensor(x, device):
    if torch.is_tensor(x):
        return x.to(device=device, dtype=F32).reshape(-1)
    return torch.tensor(list(x), device=device, dtype=F32)


def sample_reference_model(args, device, transformer, vae, encoder_hidden_states, pooled_prompt_embeds, text_ids,
                           reward_function, caption, timesteps_train, global_step, reward_weights):
    """Group rollout (reference train_grpo_flux.py:184-329) -> (rewards, all_latents [B,T'+1,N,64],
    all_log_probs [B,T'], sigma_schedule, all_image_ids [B,N,3]).  Batched over the group."""
    w, h = args.w, args.h
    T = args.sampling_steps
    sigma_schedule = SU.sd3_time_shift(args.shift, torch.linspace(1, 0, T + 1))     # host-resident
    assert len(sigma_schedule) == T + 1, "sigma_schedule must have length sample_steps + 1"
    B = encoder_hidden_states.shape[0]
    lw, lh = w // 8, h // 8
    inj = getattr(args, "injected_noise", None)      # parity tests only: {"x_T": [1,16,h,w], "steps": [T x [B,N,64]]}
    if args.init_same_noise:
        x_T = inj["x_T"].to(device) if inj else torch.randn((1, 16, lh, lw), device=device, dtype=BF16)
    if args.training_strategy == "part":
        det = [True] * T
        for i in timesteps_train:
            det[i] = False
    elif args.training_strategy == "all":
        det = [False] * T
    else:
        raise ValueError(f"training_strategy {args.training_strategy} is not supported.")
    image_ids = prepare_latent_image_ids(1, lh // 2, lw // 2, device, BF16)
    rb = args.rollout_batch if getattr(args, "rollout_batch", 0) else B
    # one prompt per rank and a shared x_T: the group's rows are identical until the first SDE step
    shared = bool(args.init_same_noise and getattr(args, "use_group", False) and B == args.num_generations
                  and getattr(args, "share_rollout_prefix", True))
    lat_chunks, lp_chunks, final_chunks = [], [], []
    for b0 in range(0, B, rb):
        nb = min(rb, B - b0)
        x0 = x_T.expand(nb, -1, -1, -1).contiguous() if args.init_same_noise else \
            torch.randn((nb, 16, lh, lw), device=device, dtype=BF16)
        z0 = pack_latents(x0, nb, 16, lh, lw)
        with torch.no_grad():
            _, latents, bl, blp = SU.run_sample_step(args, z0, range(T), sigma_schedule, transformer,
                                                     encoder_hidden_states[b0:b0 + nb],
                                                     pooled_prompt_embeds[b0:b0 + nb], text_ids[b0:b0 + 1], image_ids,
                                                     True, det,
                                                     noises=[n[b0:b0 + nb].to(device) for n in inj["steps"]] if inj else None,
                                                     shared_rows=shared)
        lat_chunks.append(bl)
        lp_chunks.append(blp)
        final_chunks.append(latents)
    all_latents = lat_chunks[0] if len(lat_chunks) == 1 else torch.cat(lat_chunks, 0)
    all_log_probs = lp_chunks[0] if len(lp_chunks) == 1 else torch.cat(lp_chunks, 0)
    final = final_chunks[0] if len(final_chunks) == 1 else torch.cat(final_chunks, 0)
    total, heads = reward_function(final, caption)
    if args.multi_reward_mix == "reward_aggr":
        rewards = _as_tensor(total, device)
    elif args.multi_reward_mix == "advantage_aggr":
        rewards = {k: _as_tensor(v, device) for k, v in heads.items()}
    else:
        raise ValueError(f"multi_reward_mix {args.multi_reward_mix} is not supported.")
    all_image_ids = image_ids.unsqueeze(0).expand(B, -1, -1)
    return rewards, all_latents, all_log_probs, sigma_schedule, all_image_ids


def compute_advantages(args, rewards, reward_weights, gathered):
    """Group-relative (or global) advantages on the device (reference train_grpo_flux.py:439-501)."""
    G = args.num_generations
    if args.use_group:
        if args.multi_reward_mix == "advantage_aggr":
            first = next(iter(rewards.values()))
            adv = torch.zeros_like(first)
            for k, r in rewards.items():
                check(lib().mgx_group_advantage(ptr(r.contiguous()), ptr(adv), r.numel(), G, float(args.trimmed_ratio),
                                                float(reward_weights[k]), 1, stream()))
            return adv
        r = rewards.contiguous()
        adv = torch.zeros_like(r)
        check(lib().mgx_group_advantage(ptr(r), ptr(adv), r.numel(), G, float(args.trimmed_ratio), 1.0, 0, stream()))
        return adv
    if args.multi_reward_mix == "advantage_aggr":
        raise ValueError("multi_reward_mix 'advantage_aggr' is not supported when use_group is False.")
    r = rewards.contiguous()
    adv = torch.empty_like(r)
    ga = gathered.contiguous()
    check(lib().mgx_global_advantage(ptr(r), ptr(ga), ptr(adv), r.numel(), ga.numel(), stream()))
    return adv


def _grad_reducer(transformer):
    """The bucketed SUM all-reduce over the model's flat gradient buffer (dist_utils.GradReducer; mode / overlap from
    `transformer.dp_grad_dtype` / `transformer.dp_overlap`, else the MGX_DP_GRAD_DTYPE / MGX_DP_OVERLAP environment)."""
    g = transformer.store.ensure_grad()
    r = getattr(transformer, "_mgx_grad_reducer", None)
    if r is None or r.flat.data_ptr() != g.data_ptr():
        r = GradReducer(g, mode=getattr(transformer, "dp_grad_dtype", None), overlap=getattr(transformer, "dp_overlap", None))
        transformer._mgx_grad_reducer = r
    return r


def _fused_step(transformer, optimizer, max_grad_norm):
    """clip_grad_norm_ + optimizer.step() (reference :606-607).  With the flat store and FusedAdamW: DP gradient
    all-reduce (sum) -> one sum-of-squares pass -> AdamW with the clip factor inside.  Returns the device grad norm."""
    ws = world_size()
    store = getattr(transformer, "store", None)
    if store is not None and hasattr(optimizer, "grad_sqnorm"):
        _grad_reducer(transformer).finish()          # (buckets launched during the backward, if any, + the rest)
        nsq = optimizer.grad_sqnorm()
        optimizer.step(max_grad_norm=max_grad_norm, grad_scale=1.0 / ws)
        return nsq.sqrt().squeeze(0) / ws
    if ws > 1:                                        # foreign model: average the per-parameter grads over ranks
        for prm in transformer.parameters():
            if prm.grad is not None:
                dist.all_reduce(prm.grad, op=dist.ReduceOp.SUM)
                prm.grad.div_(ws)
    gn = transformer.clip_grad_norm_(max_grad_norm)   # then the reference's own sequence (:606-607)
    optimizer.step()
    return gn


def train_one_step(args, device, transformer, vae, reward_function, optimizer, lr_scheduler, loader, noise_scheduler,
                   max_grad_norm, timesteps_train, global_step, reward_weights, trace=None):
    """One GRPO train step (reference train_grpo_flux.py:341-624).  Returns (total_loss, grad_norm,
    policy_total_loss, kl_total_loss, total_clip_frac, gathered_reward_res)."""
    optimizer.zero_grad()
    # second step on: the first step has shown the step's real memory peak; what the device still has free goes to kept FF
    # pre-activations (fewer GEMMs in the recompute pass, identical values: flux_backward.KEEP_FF)
    steps_done = getattr(transformer, "_mgx_train_steps", 0)
    if steps_done == 1 and hasattr(transformer, "grow_ff_keep"):
        transformer.grow_ff_keep()
    encoder_hidden_states, pooled_prompt_embeds, text_ids, caption = next(loader)
    encoder_hidden_states = encoder_hidden_states.to(device)
    pooled_prompt_embeds = pooled_prompt_embeds.to(device)
    text_ids = text_ids.to(device)
    G = args.num_generations
    if args.use_group:
        encoder_hidden_states = torch.repeat_interleave(encoder_hidden_states, G, dim=0)
        pooled_prompt_embeds = torch.repeat_interleave(pooled_prompt_embeds, G, dim=0)
        text_ids = torch.repeat_interleave(text_ids, G, dim=0)
        if isinstance(caption, str):
            caption = [caption] * G
        elif isinstance(caption, (list, tuple)):
            caption = [c for c in caption for _ in range(G)]
        else:
            raise ValueError(f"Unsupported caption type: {type(caption)}")

    reward, all_latents, all_log_probs, sigma_schedule, all_image_ids = sample_reference_model(
        args, device, transformer, vae, encoder_hidden_states, pooled_prompt_embeds, text_ids, reward_function, caption,
        timesteps_train, global_step, reward_weights)
    B = all_latents.shape[0]
    T = args.sampling_steps
    sig_host = sigma_schedule
    timestep_value = [int(s * 1000) for s in sig_host][:T]

    # ---- rewards: gather for logging / global normalisation (RCCL all_gather of [G] floats per head)
    if args.multi_reward_mix == "advantage_aggr":
        gathered = {k: gather_tensor(v) for k, v in reward.items()}
    else:
        gathered = gather_tensor(reward)
    adv = compute_advantages(args, reward, reward_weights, gathered)
    if trace is not None:
        trace["advantages"] = adv.clone()
        trace["log_probs"] = all_log_probs.clone()

    # ---- which (sample, step) pairs are replayed, in the reference's order
    n_trans = all_log_probs.shape[1] - 1                      # the last transition is dropped (:407-410)
    order = list(range(B))
    perms = None
    if args.training_strategy == "all":
        perms = torch.stack([torch.randperm(n_trans) for _ in range(B)])     # host RNG, like the reference (:504)
        nt = args.frozen_init_timesteps if args.frozen_init_timesteps > 0 else int(n_trans * args.timestep_fraction)
        if args.frozen_init_timesteps > 0:
            assert args.frozen_init_timesteps <= n_trans
        steps_of = lambda i: [int(perms[i][k]) for k in range(nt)]
    else:
        steps_of = lambda i: list(timesteps_train)
        nt = len(list(timesteps_train))
        if args.advantage_rerange_strategy != "null":
            if args.advantage_rerange_strategy not in ("random", "balance"):
                raise ValueError(f"advantage_rerange_strategy {args.advantage_rerange_strategy} is not supported.")
            adv_host = adv.detach().cpu()
            items = [{"idx": i, "advantages": adv_host[i]} for i in range(B)]
            items = balance_pos_neg(items, use_random=args.advantage_rerange_strategy == "random")
            order = [it["idx"] for it in items]
    accum = args.gradient_accumulation_steps
    denom = float(accum * nt) if nt > 0 else 1.0
    mb = getattr(args, "train_microbatch", 0) or 0
    log = torch.zeros(4, device=device, dtype=F32)            # loss, policy, kl, clip_frac sums (device)
    grad_norm = None
    lat_steps = all_latents.transpose(0, 1)                   # [T'+1, B, N, C] view (step-major storage)
    guidance = torch.tensor([3.5], device=device, dtype=BF16)
    txt_ids = text_ids[:1].repeat(encoder_hidden_states.shape[1], 1)
    img_ids = all_image_ids[0]
    transformer.train()
    for c0 in range(0, len(order), accum):
        chunk = order[c0:c0 + accum]
        pairs = [(i, t) for i in chunk for t in steps_of(i)]
        # The gradients of a leftover chunk (G % accum samples) are never applied: the reference backpropagates them and
        # the next train step's zero_grad() discards them (:360,605-609).  Default: do the same work.  With
        # `args.skip_dead_backward` only their forward runs (their losses still enter the logged averages): every
        # returned value and the weights are identical, the dead backward passes are not executed.
        dead = len(chunk) < accum and getattr(args, "skip_dead_backward", False)
        if pairs:
            pairs.sort(key=lambda p: p[1])                     # step-major: one coefficient set per contiguous slice
            for m0 in range(0, len(pairs), mb or len(pairs)):
                part = pairs[m0:m0 + (mb or len(pairs))]
                # data-parallel overlap: during the LAST micro-batch before an optimizer step a block's gradients are
                # final as soon as its backward is done -> its buckets go out while the rest of the backward runs
                last_mb = m0 + (mb or len(pairs)) >= len(pairs) and len(chunk) == accum
                hooked = False
                if last_mb and is_dist() and world_size() > 1 and getattr(transformer, "store", None) is not None:
                    red = _grad_reducer(transformer)
                    if red.overlap:
                        ranges = transformer.store.block_ranges()
                        transformer._grad_ready = lambda prefix, r_=red, rg_=ranges: r_.reduce_range(*rg_[prefix], async_op=True)
                        hooked = True
                try:
                    _replay_backward(args, transformer, part, lat_steps, all_log_probs, adv, encoder_hidden_states,
                                     pooled_prompt_embeds, txt_ids, img_ids, guidance, timestep_value, sig_host, denom, log,
                                     trace, need_grad=not dead)
                finally:
                    if hooked:
                        transformer._grad_ready = None
        if len(chunk) == accum:                                # optimizer step every `accum` samples (:605-609)
            grad_norm = _fused_step(transformer, optimizer, max_grad_norm)
            if trace is not None:
                trace.setdefault("grad_norms", []).append(grad_norm.clone())
            lr_scheduler.step()
            optimizer.zero_grad()
    allreduce_mean_vec_(log)                                   # one collective for the four logging averages
    if is_dist():
        dist.barrier()
    vals = log.tolist()                                        # the step's single device->host sync
    if args.multi_reward_mix == "advantage_aggr":
        rres = {k: v.mean().item() for k, v in gathered.items()}
    else:
        rres = gathered.mean().item()
    try:
        transformer._mgx_train_steps = steps_done + 1
    except Exception:                                           # (a transformer object that refuses attributes)
        pass
    return vals[0], (grad_norm.item() if grad_norm is not None else None), vals[1], vals[2], vals[3], rres


def _replay_backward(args, transformer, pairs, lat_steps, all_log_probs, adv, ehs, pooled, txt_ids, img_ids, guidance,
                     timestep_value, sig_host, denom, log, trace, need_grad=True):
    """Forward + backward of a batch of (sample, step) pairs; accumulates parameter grads and the logging sums."""
    dev = lat_steps.device
    idx_s = torch.tensor([p[0] for p in pairs], device=dev)
    idx_t = torch.tensor([p[1] for p in pairs], device=dev)
    x = lat_steps[idx_t, idx_s].contiguous()                   # [P, N, C] fp32 latents before the step
    nxt = lat_steps[idx_t + 1, idx_s].contiguous()             # stored next latents
    old_lp = all_log_probs[idx_s, idx_t].contiguous()
    a = adv[idx_s].contiguous()
    ts = _timestep_tensor([timestep_value[p[1]] for p in pairs], dev)
    with torch.autocast("cuda", torch.bfloat16), torch.set_grad_enabled(need_grad):
        pred = transformer(hidden_states=x, encoder_hidden_states=ehs[idx_s].contiguous(), timestep=ts,
                           guidance=guidance, txt_ids=txt_ids, pooled_projections=pooled[idx_s].contiguous(),
                           img_ids=img_ids, joint_attention_kwargs=None, return_dict=False)[0]
    P = len(pairs)
    use_flow = args.dpm_algorithm_type == "null" or ("dpmsolver" in args.dpm_algorithm_type
                                                     and args.dpm_apply_strategy == "post")
    v = pred.detach()
    new_lp = torch.empty(P, device=dev, dtype=F32)
    dv = torch.empty_like(v)
    n = x[0].numel()
    # contiguous slices of equal step index share one coefficient struct
    slices, s0 = [], 0
    for k in range(1, P + 1):
        if k == P or pairs[k][1] != pairs[s0][1]:
            slices.append((s0, k, pairs[s0][1]))
            s0 = k
    import ctypes as C
    coeffs = []
    dpm_noise = dpm_xt = None
    if not use_flow:
        # dpm_apply_strategy="all" (train_grpo_flux.py:170-180): every replay call runs a first-order SDE dpm_step from
        # a freshly default-seeded generator -- the SAME [1, N, C] noise for every (sample, step) pair -- and scores that
        # step's own sample; the stored next latent is not used.  (`injected_noise["dpm_replay"]`: parity tests only.)
        inj = getattr(args, "injected_noise", None)
        one = inj["dpm_replay"].to(dev) if inj and "dpm_replay" in inj else \
            torch.randn((1,) + tuple(x.shape[1:]), generator=torch.Generator(device=dev), device=dev, dtype=F32)
        dpm_noise = one.to(F32).expand(P, -1, -1).contiguous()
        dpm_xt = torch.empty_like(x)
    for lo, hi, t in slices:
        if not use_flow:
            kf = SU.dpm_coeffs(args.dpm_algorithm_type, args.dpm_solver_type, 1, sig_host, t, True)
            check(lib().mgx_dpm_step_fwd(ptr(x[lo:hi]), ptr(v[lo:hi]), None, None, ptr(dpm_noise[lo:hi]),
                                         dpm_xt[lo:hi].data_ptr(), None, new_lp[lo:hi].data_ptr(),
                                         ptr(SU.logp_workspace(hi - lo, n, dev)), hi - lo, n, C.byref(kf), stream()))
        elif args.flow_grpo_sampling:
            kf = SU.flow_coeffs(sig_host, t, args.eta)
            check(lib().mgx_flow_step_fwd(ptr(x[lo:hi]), ptr(v[lo:hi]), None, ptr(nxt[lo:hi]), None, None, None,
                                          new_lp[lo:hi].data_ptr(), ptr(SU.logp_workspace(hi - lo, n, dev)), hi - lo, n,
                                          C.byref(kf), 0, stream()))
        else:
            kf = SU.dance_coeffs(sig_host, t, args.eta)
            check(lib().mgx_dance_step_fwd(ptr(x[lo:hi]), ptr(v[lo:hi]), None, ptr(nxt[lo:hi]), None, None,
                                           new_lp[lo:hi].data_ptr(), ptr(SU.logp_workspace(hi - lo, n, dev)), hi - lo, n,
                                           C.byref(kf), 1, stream()))
        coeffs.append(kf)
    out = torch.empty(5, P, device=dev, dtype=F32)             # loss, policy, kl, clip_frac, g_logp
    check(lib().mgx_grpo_loss(ptr(new_lp), ptr(old_lp), ptr(a), P, float(args.clip_range), float(args.adv_clip_max),
                              float(args.kl_coeff), denom, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                              out[3].data_ptr(), out[4].data_ptr(), stream()))
    g_lp = out[4]
    for (lo, hi, t), kf in zip(slices, coeffs):
        if not use_flow:
            check(lib().mgx_dpm_step_bwd(ptr(x[lo:hi]), ptr(v[lo:hi]), ptr(dpm_xt[lo:hi]), g_lp[lo:hi].data_ptr(),
                                         dv[lo:hi].data_ptr(), hi - lo, n, C.byref(kf),
                                         SU._u(SU._host(sig_host).to(F32)[t]), stream()))
        elif args.flow_grpo_sampling:
            check(lib().mgx_flow_step_bwd(ptr(x[lo:hi]), ptr(v[lo:hi]), ptr(nxt[lo:hi]), g_lp[lo:hi].data_ptr(),
                                          dv[lo:hi].data_ptr(), hi - lo, n, C.byref(kf), stream()))
        else:
            check(lib().mgx_dance_step_bwd(ptr(x[lo:hi]), ptr(v[lo:hi]), ptr(nxt[lo:hi]), g_lp[lo:hi].data_ptr(),
                                           dv[lo:hi].data_ptr(), hi - lo, n, C.byref(kf), 1, stream()))
    if pred.requires_grad:
        pred.backward(dv)
    log += out[:4].sum(dim=1)
    if trace is not None:
        trace.setdefault("new_log_probs", []).append((list(pairs), new_lp.clone()))
        trace.setdefault("g_logp", []).append(g_lp.clone())
