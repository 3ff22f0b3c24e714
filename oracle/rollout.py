"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product.

Restates the reference's T-step mixed ODE/SDE rollout loop, pinned bit-exactly against
tests/golden/rollout.* (made by running the reference's `run_sample_step` here).

Follows /root/reference/fastvideo/utils/sampling_utils.py:12-155 (`run_sample_step`), including
the MixGRPO-Flash post-window schedule rebuild (:29-59) and the solver dispatch (:83-144).
"""
import torch

from . import solver as S


def flash_schedule(sigma_schedule, determistic, ratio, shift):
    """Post-window compression (sampling_utils.py:33-54): keep sigmas up to the last SDE step, then
    `num_post` shifted-linspace sigmas from the *unshifted* time of the following step down to 0."""
    n = sigma_schedule.size(0)
    sde_idx = [i for i, d in enumerate(determistic) if not d]
    last_sde = sde_idx[-1]
    num_post = int(max((n - 1 - last_sde) * ratio, 1))
    t0 = torch.linspace(1, 0, n)[last_sde + 1].item()
    post = S.sd3_time_shift(shift, torch.linspace(t0, 0, num_post))
    return torch.cat([sigma_schedule[:last_sde + 1], post], dim=0), last_sde


def call_model(transformer, z, ehs, pooled, text_ids, image_ids, sigma):
    """Model call contract (sampling_utils.py:63-82): timestep = int(sigma*1000)/1000, guidance 3.5 bf16."""
    tv = int(sigma * 1000)
    ts = torch.full([ehs.shape[0]], tv, dtype=torch.long)
    return transformer(hidden_states=z, encoder_hidden_states=ehs, timestep=ts / 1000,
                       guidance=torch.tensor([3.5], dtype=torch.bfloat16),
                       txt_ids=text_ids.repeat(ehs.shape[1], 1), pooled_projections=pooled, img_ids=image_ids,
                       joint_attention_kwargs=None, return_dict=False)[0]


def run_sample_step(args, z, progress_bar, sigma_schedule, transformer, encoder_hidden_states, pooled_prompt_embeds,
                    text_ids, image_ids, grpo_sample, determistic, noises=None):
    """`noises`: optional iterator of pre-drawn noise tensors (consumed in the reference's RNG order)."""
    noises = iter(noises) if noises is not None else None

    def draw(shape, dtype, generator=None):
        if noises is not None:
            return next(noises)
        return torch.randn(shape, generator=generator, dtype=dtype)

    all_latents, all_log_probs = [z], []
    use_dpm = "dpmsolver" in args.dpm_algorithm_type
    post = use_dpm and args.dpm_apply_strategy == "post"
    state = S.DPMState(order=args.dpm_solver_order) if use_dpm else None
    last_sde = None
    if post:
        assert args.sample_strategy == "progressive"
        sigma_schedule, last_sde = flash_schedule(sigma_schedule, determistic, args.dpm_post_compress_ratio, args.shift)
        progress_bar = range(sigma_schedule.size(0) - 1)

    x0 = None
    for i in progress_bar:
        transformer.eval()
        pred = call_model(transformer, z, encoder_hidden_states, pooled_prompt_embeds, text_ids, image_ids,
                          sigma_schedule[i])
        zf = z.to(torch.float32)
        flow_like = (not use_dpm) or (post and i <= last_sde)
        if flow_like:
            if args.flow_grpo_sampling:
                if post:
                    state.update(S.convert_model_output(pred, zf, sigma_schedule, i))
                nz = draw(pred.shape, pred.dtype)          # drawn on ODE steps too
                z, x0, lp, _, _ = S.flow_grpo_step(pred, zf, args.eta, sigma_schedule, i, None,
                                                   determistic=determistic[i], noise=nz)
                if post:
                    state.update_lower_order()
            else:
                sde = not determistic[i]
                nz = draw(zf.shape, torch.float32) if sde else None
                z, x0, lp = S.dance_grpo_step(pred, zf, args.eta, sigma_schedule, i, None, True, sde, noise=nz)
        elif post:
            z, x0, lp = S.dpm_step(args, pred, zf, i, sigma_schedule[:-1], sigma_schedule, dpm_state=state,
                                   sde_solver=False)
        else:  # strategy "all": a fresh default-seeded generator every step (same noise each time; App. B)
            sde = not determistic[i]
            nz = draw(zf.shape, torch.float32, torch.Generator()) if sde else None
            z, x0, lp = S.dpm_step(args, pred, zf, i, sigma_schedule[:-1], sigma_schedule, dpm_state=state,
                                   variance_noise=nz, sde_solver=sde)
        all_latents.append(z)
        all_log_probs.append(lp)

    latents = x0 if args.drop_last_sample else z.to(x0.dtype)
    return z, latents, torch.stack(all_latents, dim=1), torch.stack(all_log_probs, dim=1)
