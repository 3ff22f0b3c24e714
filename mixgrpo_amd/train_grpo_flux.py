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
        optimizer.step(max_grad_norm=max_grad_norm, grad_scale=1.0 / ws, gnorm_sq=nsq)   # (the one pass over the gradients: 8 ms)
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


# ------------------------------------------------------------------------------------------------ entry point
# `torchrun ... -m mixgrpo_amd.train_grpo_flux <the flags of scripts/finetune/finetune_flux_grpo_MixGRPO.sh:120-196>`:
# the reference's command-line surface (fastvideo/train_grpo_flux.py:894-1423) and its main loop (:627-892).

# (flag, type, default, choices): every flag the reference's parser defines; SURVEY.md Appendix D says which are read on the hot
# path, which only in main(), and which this fork defines but never reads (accepted and ignored here as well)
_VALUE_FLAGS = (
    ("data_json_path", str, None, None), ("dataloader_num_workers", int, 10, None), ("train_batch_size", int, 16, None),
    ("num_latent_t", int, 1, None), ("pretrained_model_name_or_path", str, None, None),
    ("dit_model_name_or_path", str, None, None), ("vae_model_path", str, None, None), ("cache_dir", str, "./cache_dir", None),
    ("ema_decay", float, 0.995, None), ("ema_start_step", int, 0, None), ("cfg", float, 0.0, None), ("seed", int, None, None),
    ("output_dir", str, None, None), ("checkpointing_steps", int, 500, None), ("resume_from_checkpoint", str, None, None),
    ("logging_dir", str, "logs", None), ("max_train_steps", int, None, None), ("gradient_accumulation_steps", int, 1, None),
    ("learning_rate", float, 1e-4, None), ("lr_warmup_steps", int, 10, None), ("max_grad_norm", float, 2.0, None),
    ("selective_checkpointing", float, 1.0, None), ("mixed_precision", str, None, ("no", "fp16", "bf16")),
    ("sp_size", int, 1, None), ("train_sp_batch_size", int, 1, None), ("fsdp_sharding_startegy", str, "full", None),
    ("lr_scheduler", str, "constant_with_warmup", None), ("lr_num_cycles", int, 1, None), ("lr_power", float, 1.0, None),
    ("weight_decay", float, 0.01, None), ("master_weight_type", str, "fp32", None), ("h", int, None, None),
    ("w", int, None, None), ("t", int, None, None), ("sampling_steps", int, None, None), ("eta", float, None, None),
    ("sampler_seed", int, None, None), ("loss_coef", float, 1.0, None), ("num_generations", int, 16, None),
    ("shift", float, 1.0, None), ("timestep_fraction", float, 1.0, None), ("clip_range", float, 1e-4, None),
    ("adv_clip_max", float, 5.0, None), ("advantage_rerange_strategy", str, "null", ("random", "balance", "null")),
    ("trimmed_ratio", float, 0.0, None), ("experiment_name", str, "test", None),
    ("training_strategy", str, "all", ("part", "all")), ("frozen_init_timesteps", int, -1, None), ("kl_coeff", float, 0.01, None),
    ("iters_per_group", int, 25, None), ("group_size", int, 4, None),
    ("sample_strategy", str, "progressive", ("progressive", "random", "decay", "exp_decay")), ("prog_overlap_step", int, 1, None),
    ("max_iters_per_group", int, 10, None), ("min_iters_per_group", int, 1, None),
    ("reward_model", str, "hpsv2", ("hpsv2", "clip_score", "image_reward", "pick_score", "unified_reward", "hpsv2_clip_score",
                                    "multi_reward")),
    ("hps_path", str, "hps_ckpt/HPS_v2.1_compressed.pt", None), ("hps_clip_path", str, "hps_ckpt/open_clip_pytorch_model.bin", None),
    ("clip_score_path", str, "hf-hub:apple/DFN5B-CLIP-ViT-H-14-384", None),
    ("image_reward_path", str, "./image_reward_ckpt/ImageReward.pt", None),
    ("image_reward_med_config", str, "./image_reward_ckpt/med_config.json", None), ("image_reward_http_proxy", str, None, None),
    ("image_reward_https_proxy", str, None, None), ("pick_score_http_proxy", str, None, None),
    ("pick_score_https_proxy", str, None, None), ("unified_reward_url", str, None, None),
    ("unified_reward_default_question_type", str, None, None), ("unified_reward_num_workers", int, 1, None),
    ("multi_reward_mix", str, "advantage_aggr", ("advantage_aggr", "reward_aggr")), ("hps_weight", float, 1.0, None),
    ("clip_score_weight", float, 1.0, None), ("image_reward_weight", float, 1.0, None), ("pick_score_weight", float, 1.0, None),
    ("unified_reward_weight", float, 1.0, None), ("dpm_algorithm_type", str, "null", ("null", "dpmsolver", "dpmsolver++")),
    ("dpm_apply_strategy", str, "post", ("post", "all")), ("dpm_post_compress_ratio", float, 0.4, None),
    ("dpm_solver_order", int, 2, (1, 2, 3)), ("dpm_solver_type", str, "heun", ("heun", "midpoint")), ("wandb_key", str, None, None),
)
_SWITCHES = ("precondition_outputs", "gradient_checkpointing", "allow_tf32", "use_cpu_offload", "use_group", "ignore_last",
             "init_same_noise", "flow_grpo_sampling", "drop_last_sample", "prog_overlap", "roll_back")
# engine options the reference has no flag for (all optional)
_ENGINE_FLAGS = (("rollout_batch", int, 0), ("train_microbatch", int, 0), ("attention_dtype", str, "bf16"),
                 ("mgx_max_epochs", int, None),
                 # "package.module:factory" -- factory(args) -> {RewardClassName: callable(images, prompts) -> scores}: the
                 # reward models of the decode + reward stage (their weights are not part of this engine; the reference builds
                 # its own from HF-hub names at fastvideo/train_grpo_flux.py:639-651)
                 ("mgx_reward_plugin", str, None))


def build_parser():
    import argparse
    p = argparse.ArgumentParser(description="MixGRPO FLUX trainer on the MI355X-native engine (reference CLI surface)")
    for name, typ, default, choices in _VALUE_FLAGS:
        kw = dict(type=typ, default=default)
        if choices is not None:
            kw["choices"] = list(choices)
        if name == "data_json_path":
            kw["required"] = True
        p.add_argument("--" + name, **kw)
    for name in _SWITCHES:
        p.add_argument("--" + name, action="store_true", default=False)
    for name, typ, default in _ENGINE_FLAGS:
        p.add_argument("--" + name, type=typ, default=default)
    p.add_argument("--skip_dead_backward", action="store_true", default=False)
    return p


def set_seed(seed):
    """accelerate.utils.set_seed (reference :656 `set_seed(args.seed + rank)`)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def _device_loader(dataloader, device):
    """fastvideo/utils/communications_flux.py `sp_parallel_dataloader_wrapper` at sp_size 1: the dataloader, forever, with
    the tensors moved to the device."""
    while True:
        for item in dataloader:
            ehs, pooled, text_ids, caption = item
            yield ehs.to(device), pooled.to(device), text_ids.to(device), caption


def synthetic_reward_function(heads=("SyntheticReward",), seed=1234):
    """Stand-in for decode + score when no reward function is handed to `main` (the reference's reward models need weights
    that are not available offline; this fork's own `compute_reward` is broken): a deterministic
    pseudo-reward per sample from the final latents, so that runs are reproducible and advantages are non-trivial."""
    def fn(latents, captions):
        flat = latents.detach().float().reshape(latents.shape[0], -1)
        per = {}
        for j, h in enumerate(heads):
            w = torch.sin(torch.arange(flat.shape[1], device=flat.device, dtype=torch.float32) * (0.37 + 0.11 * j) + seed)
            per[h] = torch.sigmoid((flat * w).mean(dim=1) * 40.0)
        return sum(per.values()), per
    return fn


_REWARD_CLASSES = ("HPSClipRewardModel", "CLIPScoreRewardModel", "ImageRewardModel", "PickScoreRewardModel", "UnifiedRewardModel")


def reward_weights_from_args(args):
    """Upstream keys `reward_weights` by reward CLASS name for the models `--reward_model` activates (eval_reward.py:185,224)."""
    table = {"hpsv2": ("HPSClipRewardModel",), "clip_score": ("CLIPScoreRewardModel",), "image_reward": ("ImageRewardModel",),
             "pick_score": ("PickScoreRewardModel",), "unified_reward": ("UnifiedRewardModel",),
             "hpsv2_clip_score": ("HPSClipRewardModel", "CLIPScoreRewardModel"),
             "multi_reward": ("HPSClipRewardModel", "ImageRewardModel", "PickScoreRewardModel")}
    flag = {"HPSClipRewardModel": "hps_weight", "CLIPScoreRewardModel": "clip_score_weight", "ImageRewardModel": "image_reward_weight",
            "PickScoreRewardModel": "pick_score_weight", "UnifiedRewardModel": "unified_reward_weight"}
    return {name: float(getattr(args, flag[name])) for name in table[args.reward_model]}


def main(args, reward_function=None, reward_weights=None, reward_models=None):
    """The reference's main() (:627-892) on this engine: replica data parallelism instead of FSDP (the flags that configure
    FSDP / activation checkpointing / sequence parallelism are accepted and have no effect), a JSON log line per step instead of
    wandb, and a `--resume_from_checkpoint` that actually resumes (weights, AdamW moments, LR position, SDE-window state, epoch,
    position inside the epoch, global step, data position and every rank's generator states), from a checkpoint of any epoch."""
    import json
    import os
    import time
    from collections import deque

    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler

    from .checkpoint import (load_resume_position, load_resume_state, load_rng_state, save_checkpoint, save_resume_state,
                             save_rng_state)
    from .flux import FluxTransformer2DModel
    from .grpo_states import GRPOTrainingStates
    from .latent_flux_rl_datasets import LatentDataset, latent_collate_function
    from .optim import FusedAdamW, get_scheduler

    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    rk = int(os.environ.get("RANK", 0))
    ws = int(os.environ.get("WORLD_SIZE", 1))
    if ws > 1 and not dist.is_initialized():
        dist.init_process_group("nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if args.sp_size != 1:
        raise ValueError("sp_size must be 1 for FLUX (the reference never uses sequence parallelism on this path)")
    if args.seed is not None:
        set_seed(args.seed + rk)
    run_dir = None
    if args.output_dir is not None:
        run_dir = f"{args.output_dir}/{args.training_strategy}_{args.experiment_name}"
        if rk <= 0:
            os.makedirs(run_dir, exist_ok=True)
            with open(os.path.join(run_dir, "args.json"), "w") as f:
                json.dump({k: v for k, v in vars(args).items()}, f, indent=4, default=str)

    model_path = args.resume_from_checkpoint or args.pretrained_model_name_or_path
    main_print(f"--> loading model from {model_path}")
    transformer = FluxTransformer2DModel.from_pretrained(model_path, device=device, subfolder="transformer", torch_dtype=F32)
    transformer.attention_dtype = args.attention_dtype
    transformer.train()
    optimizer = FusedAdamW(transformer, lr=args.learning_rate, betas=(0.9, 0.999), weight_decay=args.weight_decay, eps=1e-8)
    # the reference's call (:726-734); an unknown name fails here with the list of supported ones
    lr_scheduler = get_scheduler(args.lr_scheduler, optimizer=optimizer, num_warmup_steps=args.lr_warmup_steps,
                                 num_training_steps=1000000, num_cycles=args.lr_num_cycles, power=args.lr_power, last_epoch=-1)
    # the decode stage of the rollout (:697-701): `AutoencoderKL.from_pretrained(<pretrained>, subfolder="vae", bf16)` when
    # the model directory holds one (the HIP decode of mixgrpo_amd/vae.py)
    vae = None
    if os.path.isdir(os.path.join(args.pretrained_model_name_or_path, "vae")):
        from .vae import AutoencoderKL
        vae = AutoencoderKL.from_pretrained(args.pretrained_model_name_or_path, subfolder="vae", torch_dtype=BF16, device=device)
        main_print(f"--> VAE loaded from {args.pretrained_model_name_or_path}/vae")

    train_dataset = LatentDataset(args.data_json_path, args.num_latent_t, args.cfg)
    sampler = DistributedSampler(train_dataset, rank=rk, num_replicas=ws, shuffle=True, seed=args.sampler_seed or 0)
    train_dataloader = DataLoader(train_dataset, sampler=sampler, collate_fn=latent_collate_function, pin_memory=True,
                                  batch_size=args.train_batch_size, num_workers=args.dataloader_num_workers, drop_last=True)
    loader = _device_loader(train_dataloader, device)

    if reward_function is None and reward_models:
        # decode + score, as the reference's rollout does per sample (:279-316): reward models keyed by reward CLASS name
        # (callables `(images, prompts) -> scores`; their weights are not available offline, the caller hands them in)
        if vae is None:
            raise ValueError(f"reward models need decoded images, but {args.pretrained_model_name_or_path}/vae does not exist")
        from .reward_adapter import make_reward_function
        if reward_weights is None:
            reward_weights = {k: 1.0 for k in reward_models} if any(k not in _REWARD_CLASSES for k in reward_models) \
                else {k: v for k, v in reward_weights_from_args(args).items() if k in reward_models}
        reward_function = make_reward_function(vae, reward_models, reward_weights, args.h, args.w)
        main_print(f"--> rewards: HIP VAE decode + {', '.join(reward_models)}")
    if reward_function is None:
        heads = ("SyntheticReward",)
        reward_function = synthetic_reward_function(heads)
        reward_weights = {h: 1.0 for h in heads}
        main_print("--> no reward function given: synthetic rewards (decode + reward models are outside this engine)")
    elif reward_weights is None:
        reward_weights = reward_weights_from_args(args)

    grpo_states = None
    if args.training_strategy == "part":
        grpo_states = GRPOTrainingStates(
            iters_per_group=args.iters_per_group, group_size=args.group_size, max_timesteps=args.sampling_steps - 2,
            cur_timestep=0, cur_iter_in_group=0, sample_strategy=args.sample_strategy, prog_overlap=args.prog_overlap,
            prog_overlap_step=args.prog_overlap_step, max_iters_per_group=args.max_iters_per_group,
            min_iters_per_group=args.min_iters_per_group, roll_back=args.roll_back)

    init_steps, start_epoch, steps_done = 0, 0, 0
    if args.resume_from_checkpoint:
        init_steps = load_resume_state(args.resume_from_checkpoint, optimizer, lr_scheduler, grpo_states)
        start_epoch, _, steps_done = load_resume_position(args.resume_from_checkpoint)
        # the prompts the finished steps consumed (one batch per train step), drawn exactly as the uninterrupted run drew
        # them: every finished epoch's `set_epoch` before its steps, so the dataloader reshuffles at the same positions
        for e in range(start_epoch + 1):
            sampler.set_epoch(e)
            for _ in range(args.max_train_steps if e < start_epoch else init_steps):
                next(loader)
        load_rng_state(args.resume_from_checkpoint, rk)  # last: nothing below draws from a generator before the next step does
        main_print(f"--> resumed from {args.resume_from_checkpoint} at epoch {start_epoch}, step {init_steps} "
                   f"({steps_done} train steps done)")
    main_print("***** Running training *****")
    main_print(f"  Num examples = {len(train_dataset)}  world size = {ws}  resume step = {init_steps}")
    main_print(f"  Gradient Accumulation steps = {args.gradient_accumulation_steps}  steps per epoch = {args.max_train_steps}")

    step_times = deque(maxlen=100)
    resumed, first_step = bool(args.resume_from_checkpoint), init_steps + 1      # (the resumed-from checkpoint is not rewritten)
    global_step = steps_done - 1
    log_path = os.path.join(run_dir, "train_log.jsonl") if run_dir and rk <= 0 else None
    epochs = 1000000 if args.mgx_max_epochs is None else args.mgx_max_epochs
    for epoch in range(start_epoch, epochs):
        sampler.set_epoch(epoch)
        for step in range(init_steps + 1, args.max_train_steps + 1):
            global_step += 1
            start_time = time.time()
            if step % args.checkpointing_steps == 0 and run_dir is not None and \
                    not (resumed and step == first_step and epoch == start_epoch):
                d = save_checkpoint(transformer, rk, run_dir, step, epoch)
                # (what the NEXT step needs to continue: it is `step` itself that has not run yet)
                save_resume_state(d, optimizer, lr_scheduler, grpo_states, global_step=step - 1, rank=rk, epoch=epoch,
                                  steps_done=global_step)
                if ws > 1:
                    dist.barrier()                      # the directory exists before the other ranks write into it
                save_rng_state(d, rk)
                if ws > 1:
                    dist.barrier()
            if args.training_strategy == "part":
                timesteps_train = grpo_states.get_current_timesteps()
                grpo_states.update_iteration()
            else:
                timesteps_train = list(range(args.sampling_steps))
            loss, grad_norm, policy_loss, kl_loss, clip_frac, reward = train_one_step(
                args, device, transformer, vae, reward_function, optimizer, lr_scheduler, loader, None, args.max_grad_norm,
                timesteps_train, global_step, reward_weights)
            step_time = time.time() - start_time
            step_times.append(step_time)
            if rk <= 0:
                log = {"step": step, "global_step": global_step, "epoch": epoch, "train_loss": loss, "policy_loss": policy_loss,
                       "kl_loss": kl_loss, "clip_frac": clip_frac, "grad_norm": grad_norm, "timesteps_train": list(timesteps_train),
                       "cur_timesteps": grpo_states.cur_timestep if grpo_states else 0,
                       "cur_iter_in_group": grpo_states.cur_iter_in_group if grpo_states else 0,
                       "learning_rate": lr_scheduler.get_last_lr()[0], "step_time": step_time,
                       "avg_step_time": sum(step_times) / len(step_times)}
                if args.multi_reward_mix == "advantage_aggr" and isinstance(reward, dict):
                    for name, val in reward.items():
                        log[f"reward_{name}"] = val
                else:
                    log["reward"] = reward
                line = json.dumps(log, default=lambda x: x.item() if hasattr(x, "item") else str(x))
                print(line, flush=True)
                if log_path:
                    with open(log_path, "a") as f:
                        f.write(line + "\n")
        init_steps = 0
    if ws > 1 and dist.is_initialized():
        dist.barrier()
    return transformer


def load_reward_plugin(spec, args):
    """`--mgx_reward_plugin package.module:factory` -> the factory's {RewardClassName: callable} (None when no plugin is named)."""
    if not spec:
        return None
    import importlib
    mod, _, fn = spec.partition(":")
    if not mod or not fn:
        raise ValueError(f"--mgx_reward_plugin {spec!r}: expected 'package.module:factory'")
    models = getattr(importlib.import_module(mod), fn)(args)
    if not isinstance(models, dict) or not models or not all(callable(v) for v in models.values()):
        raise ValueError(f"--mgx_reward_plugin {spec!r}: the factory must return a non-empty {{name: callable}} dict")
    return models


if __name__ == "__main__":
    _args = build_parser().parse_args()
    main(_args, reward_models=load_reward_plugin(_args.mgx_reward_plugin, _args))
