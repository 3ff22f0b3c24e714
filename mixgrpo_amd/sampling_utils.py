"""MI355X-native sampler surface: drop-in for `fastvideo.utils.sampling_utils` (reference
fastvideo/utils/sampling_utils.py) on the hot path.  Same function names, argument meaning and
error behaviour; the arithmetic runs in hand-written HIP kernels (csrc/solver.hip) through the C ABI.

Host side (this file) only: (1) per-step scalar coefficients, computed ONCE per step on the host from
the (CPU-resident) sigma schedule in the reference's fp32 operation order -- no `.item()` device sync
in the loop; (2) output allocation; (3) the rollout loop and its DPM-Solver state.

`SCALAR_ROUNDING`: PyTorch rounds a 0-dim fp32 scalar to bf16 before multiplying a bf16 tensor when the
product is spelled `scalar * tensor` (always, on a device), but not for `tensor * scalar` on CPU.  "cpu"
(default) reproduces the golden vectors taken from the reference on CPU; "device" rounds every such scalar,
which is what the reference's eager ops do with device-resident 0-dim sigmas (numerics).
"""
import ctypes as C
import math
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch

from . import _lib
from ._lib import DanceCoeffs, DpmCoeffs, FlowCoeffs, check, lib, logp_workspace, ptr, stream

SCALAR_ROUNDING = "cpu"
_F32 = torch.float32
_BF16 = torch.bfloat16
_LOG_SQRT_2PI = torch.log(torch.sqrt(2 * torch.as_tensor(math.pi)))


def sd3_time_shift(shift, t):
    """reference sampling_utils.py:9-10"""
    return (shift * t) / (1 + (shift - 1) * t)


def _host(sigmas):
    """The schedule as a CPU fp32 tensor (a device schedule is copied once; keep it on the host upstream)."""
    return sigmas if not sigmas.is_cuda else sigmas.detach().cpu()


def _r(c):
    """0-dim fp32 -> python float after the bf16 rounding PyTorch applies to `scalar * bf16_tensor`."""
    return c.to(_BF16).to(_F32).item()


def _u(c, tensor_first=True):
    """Scalar of a `tensor * scalar` product: full precision on CPU semantics, bf16 on device semantics."""
    return _r(c) if SCALAR_ROUNDING == "device" else c.item()


def flow_coeffs(sigmas, index, eta) -> FlowCoeffs:
    """Scalars of flow_grpo_step (reference sampling_utils.py:170-177,186,195,199,202-204)."""
    sg = _host(sigmas).to(_F32)
    s = sg[index]
    dt = sg[index + 1] - s
    s_den = torch.where(s == 1, sg[1], s)
    std = torch.sqrt(s / (1 - s_den)) * eta
    c_x = 1 + std ** 2 / (2 * s) * dt
    c_v = 1 + std ** 2 * (1 - s) / (2 * s)
    sd = std * torch.sqrt(-1 * dt)
    k = FlowCoeffs(sigma_x0=_r(s), c_x=c_x.item(), c_v=_u(c_v), dt_mean=_u(dt), sd_noise=_r(sd), dt_det=_r(dt),
                   den=(2 * (sd ** 2)).item(), log_sd=torch.log(sd).item(), log_c=_LOG_SQRT_2PI.item())
    k.sd_value = sd.item()
    return k


def dance_coeffs(sigmas, index, eta) -> DanceCoeffs:
    """Scalars of dance_grpo_step (reference sampling_utils.py:222-234,245)."""
    sg = _host(sigmas).to(_F32)
    s = sg[index]
    ds = sg[index + 1] - s
    sd = eta * math.sqrt(s - sg[index + 1])
    f32 = lambda x: torch.tensor(x, dtype=_F32).item()  # python scalar as PyTorch casts it for an fp32 tensor op
    return DanceCoeffs(ds_r=_r(ds), s_r=_r(s), ds=ds.item(), ds_b=_u(ds), s_b=_u(s), one_m_s=(1 - s).item(),
                       s_sq=(s ** 2).item(), half_eta2=f32(-0.5 * eta ** 2), sd=f32(sd), den=f32(2 * (sd ** 2)))


def _lam(sig):
    return torch.log(1 - sig) - torch.log(sig)


def dpm_coeffs(algo, stype, order, sigmas, i, sde) -> DpmCoeffs:
    """Scalars of the DPM-Solver(++) multistep updates (reference sampling_utils.py:398-639), signs folded in."""
    sg = _host(sigmas).to(_F32)
    sig_t, sig_s0 = sg[i + 1], sg[i]
    a_t, a_s0 = 1 - sig_t, 1 - sig_s0
    h = _lam(sig_t) - _lam(sig_s0)
    k = DpmCoeffs(order=order, sde=int(bool(sde)), sigma_x0=_r(sig_s0))
    zero = torch.zeros((), dtype=_F32)
    r0 = r1 = None
    if order >= 2:
        lam_s1 = _lam(sg[i - 1])
        r0 = (_lam(sig_s0) - lam_s1) / h
        k.inv_r0 = (1.0 / r0).item()
    if order == 3:
        r1 = (lam_s1 - _lam(sg[i - 2])) / h
        k.inv_r1 = (1.0 / r1).item()
        k.c_r = (r0 / (r0 + r1)).item()
        k.inv_r01 = (1.0 / (r0 + r1)).item()
    cm, cx = [zero] * 4, [zero] * 4
    if algo == "dpmsolver++":
        e2 = 1 - torch.exp(-2.0 * h)
        em1 = torch.exp(-h) - 1.0
        cm[0], cm[1] = sig_t / sig_s0 * torch.exp(-h), a_t * e2
        cx[0], cx[1] = sig_t / sig_s0, -(a_t * em1)
        if order == 2 and stype == "midpoint":
            cm[2], cx[2] = 0.5 * (a_t * e2), -(0.5 * (a_t * em1))
        elif order >= 2:
            cm[2] = a_t * ((1.0 - torch.exp(-2.0 * h)) / (-2.0 * h) + 1.0)
            cx[2] = a_t * (em1 / h + 1.0)
        if order == 3:
            cm[3] = a_t * ((1.0 - torch.exp(-2.0 * h) - 2.0 * h) / (2.0 * h) ** 2 - 0.5)
            cx[3] = -(a_t * ((torch.exp(-h) - 1.0 + h) / h ** 2 - 0.5))
        dt_sqrt = torch.sqrt(1.0 - torch.exp(-2 * h))
    elif algo == "dpmsolver":
        if order == 3:
            raise NotImplementedError("third-order 'dpmsolver' is unreachable in the reference "
                                      "(sampling_utils.py:629-639 returns an unbound prev_mean)")
        eh1 = torch.exp(h) - 1.0
        cm[0], cm[1] = a_t / a_s0, -(2.0 * (sig_t * eh1))
        cx[0], cx[1] = a_t / a_s0, -(sig_t * eh1)
        if order == 2 and stype == "midpoint":
            cm[2], cx[2] = -(sig_t * eh1), -(0.5 * (sig_t * eh1))
        elif order == 2:
            cm[2], cx[2] = -(2.0 * (sig_t * (eh1 / h - 1.0))), -(sig_t * (eh1 / h - 1.0))
        dt_sqrt = torch.sqrt(torch.exp(2 * h) - 1.0)
    else:
        raise ValueError(f"unknown dpm_algorithm_type {algo!r}")
    sdn = sig_t * dt_sqrt
    for j in range(4):
        k.cm[j] = cm[j].item()
        k.cx[j] = cx[j].item()
    k.sd_noise = sdn.item()
    k.den = (2 * (sdn ** 2)).item()
    k.log_sd = torch.log(sdn).item()
    k.log_c = _LOG_SQRT_2PI.item()
    return k


def _flat(t):
    return t.shape[0], t[0].numel()


def _as(t, dtype):
    t = t.detach()
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


class _FlowReplayLogp(torch.autograd.Function):
    """log-prob of a stored transition with d logp / d model_output (reference train_grpo_flux.py:149-157)."""

    @staticmethod
    def forward(ctx, model_output, latents, prev_sample, k):
        B, n = _flat(latents)
        v = _as(model_output, _BF16)
        logp = torch.empty(B, dtype=_F32, device=v.device)
        check(lib().mgx_flow_step_fwd(ptr(latents), ptr(v), None, ptr(prev_sample), None, None, None, ptr(logp),
                                      ptr(logp_workspace(B, n, v.device)), B, n, C.byref(k), 0, stream()))
        ctx.save_for_backward(v, latents, prev_sample)
        ctx.k = k
        return logp

    @staticmethod
    def backward(ctx, g):
        v, latents, prev_sample = ctx.saved_tensors
        B, n = _flat(latents)
        dv = torch.empty_like(v)
        check(lib().mgx_flow_step_bwd(ptr(latents), ptr(v), ptr(prev_sample), ptr(_as(g, _F32)), ptr(dv), B, n,
                                      C.byref(ctx.k), stream()))
        return dv, None, None, None


class _DanceReplayLogp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model_output, latents, prev_sample, k, sde):
        B, n = _flat(latents)
        v = _as(model_output, _BF16)
        logp = torch.empty(B, dtype=_F32, device=v.device)
        check(lib().mgx_dance_step_fwd(ptr(latents), ptr(v), None, ptr(prev_sample), None, None, ptr(logp),
                                       ptr(logp_workspace(B, n, v.device)), B, n, C.byref(k), int(sde), stream()))
        ctx.save_for_backward(v, latents, prev_sample)
        ctx.k, ctx.sde = k, sde
        return logp

    @staticmethod
    def backward(ctx, g):
        v, latents, prev_sample = ctx.saved_tensors
        B, n = _flat(latents)
        dv = torch.empty_like(v)
        check(lib().mgx_dance_step_bwd(ptr(latents), ptr(v), ptr(prev_sample), ptr(_as(g, _F32)), ptr(dv), B, n,
                                       C.byref(ctx.k), int(ctx.sde), stream()))
        return dv, None, None, None, None


class _DpmSdeLogp(torch.autograd.Function):
    """First-order SDE dpm_step (no multistep state) -> (x_t, x0, log_prob) with d log_prob / d model_output: the replay of
    reference train_grpo_flux.py:170-180, whose log-prob differentiates through prev_sample_mean with the drawn sample
    detached (sampling_utils.py:376-383)."""

    @staticmethod
    def forward(ctx, model_output, x, noise, k, sigma_b, out):
        B, n = _flat(x)
        v = _as(model_output, _BF16)
        x0 = torch.empty_like(x)
        logp = torch.empty(B, dtype=_F32, device=v.device)
        check(lib().mgx_dpm_step_fwd(ptr(x), ptr(v), None, None, ptr(noise), ptr(out), ptr(x0), ptr(logp),
                                     ptr(logp_workspace(B, n, v.device)), B, n, C.byref(k), stream()))
        ctx.save_for_backward(v, x, out)
        ctx.k, ctx.sigma_b = k, sigma_b
        ctx.mark_non_differentiable(out, x0)
        return out, x0, logp

    @staticmethod
    def backward(ctx, g_out, g_x0, g):
        v, x, xt = ctx.saved_tensors
        B, n = _flat(x)
        dv = torch.empty_like(v)
        check(lib().mgx_dpm_step_bwd(ptr(x), ptr(v), ptr(xt), ptr(_as(g, _F32)), ptr(dv), B, n, C.byref(ctx.k),
                                     ctx.sigma_b, stream()))
        return dv, None, None, None, None, None


def flow_grpo_step(model_output, latents, eta, sigmas, index, prev_sample, generator=None, determistic=False,
                   noise=None, prev_out=None, want_x0=True, want_mean=True):
    """Flow-GRPO SDE Euler-Maruyama step + Gaussian log-prob (reference sampling_utils.py:157-210).

    Returns (prev_sample fp32, pred_original_sample fp32, log_prob fp32 [B], prev_sample_mean fp32,
    std_dev_t*sqrt(-dt) 0-dim).  Extras over the reference signature (all optional): `noise` injects the
    draw, `prev_out` is a preallocated output, `want_x0/want_mean=False` skips those outputs (None returned)
    to keep the step at its algorithmic 12 B/element.
    """
    if prev_sample is not None and generator is not None:
        raise ValueError("Cannot pass both generator and prev_sample. Please make sure that either `generator` or"
                         " `prev_sample` stays `None`.")
    index = int(index)
    k = flow_coeffs(sigmas, index, eta)
    dev = model_output.device
    x = _as(latents, _F32)
    B, n = _flat(x)
    sd_t = torch.tensor(k.sd_value, dtype=_F32, device=dev)
    if prev_sample is not None:
        prev = _as(prev_sample, _F32)
        if model_output.requires_grad and torch.is_grad_enabled():
            if determistic:
                raise NotImplementedError("deterministic replay with autograd is not on the reference's path")
            logp = _FlowReplayLogp.apply(model_output, x, prev, k)
            x0 = mean = None
            if want_x0 or want_mean:  # off the hot path: recomputed without grad
                with torch.no_grad():
                    _, x0, _, mean, _ = flow_grpo_step(model_output.detach(), x, eta, sigmas, index, prev,
                                                       want_x0=want_x0, want_mean=want_mean)
            return prev, x0, logp, mean, sd_t
        v = _as(model_output, _BF16)
        x0 = torch.empty_like(x) if want_x0 else None
        mean = torch.empty_like(x) if want_mean else None
        logp = torch.empty(B, dtype=_F32, device=dev)
        check(lib().mgx_flow_step_fwd(ptr(x), ptr(v), None, ptr(prev), None, ptr(x0), ptr(mean), ptr(logp),
                                      ptr(logp_workspace(B, n, dev)), B, n, C.byref(k), 0, stream()))
        if determistic:  # the reference overrides prev_sample even when one was passed (:198-199)
            out = _ode_override(x, v, k)
            # and the log-prob is then taken at the overridden sample
            check(lib().mgx_flow_step_fwd(ptr(x), ptr(v), None, ptr(out), None, None, None, ptr(logp),
                                          ptr(logp_workspace(B, n, dev)), B, n, C.byref(k), 0, stream()))
            return out, x0, logp, mean, sd_t
        return prev, x0, logp, mean, sd_t
    v = _as(model_output, _BF16)
    if noise is None:
        noise = torch.randn(model_output.shape, generator=generator, device=dev, dtype=model_output.dtype)
    nz = _as(noise, _BF16)
    out = prev_out if prev_out is not None else torch.empty_like(x)
    x0 = torch.empty_like(x) if want_x0 else None
    mean = torch.empty_like(x) if want_mean else None
    logp = torch.empty(B, dtype=_F32, device=dev)
    check(lib().mgx_flow_step_fwd(ptr(x), ptr(v), ptr(nz), None, ptr(out), ptr(x0), ptr(mean), ptr(logp),
                                  ptr(logp_workspace(B, n, dev)), B, n, C.byref(k), int(bool(determistic)), stream()))
    return out, x0, logp, mean, sd_t


def _ode_override(x, v, k):
    """prev = x + dt*v via the deterministic branch of the forward kernel (noise ignored)."""
    B, n = _flat(x)
    out = torch.empty_like(x)
    logp = torch.empty(B, dtype=_F32, device=x.device)
    v = _as(v, _BF16)
    check(lib().mgx_flow_step_fwd(ptr(x), ptr(v), ptr(v), None, ptr(out), None, None, ptr(logp),
                                  ptr(logp_workspace(B, n, x.device)), B, n, C.byref(k), 1, stream()))
    return out


def dance_grpo_step(model_output, latents, eta, sigmas, index, prev_sample, grpo, sde_solver, noise=None,
                    prev_out=None):
    """DanceGRPO SDE/ODE step (reference sampling_utils.py:212-253).  grpo=True -> (prev, x0, log_prob);
    grpo=False -> (prev_sample_mean, x0)."""
    index = int(index)
    k = dance_coeffs(sigmas, index, eta)
    dev = model_output.device
    x = _as(latents, _F32)
    B, n = _flat(x)
    if grpo and prev_sample is not None and model_output.requires_grad and torch.is_grad_enabled():
        prev = _as(prev_sample, _F32)
        return prev, None, _DanceReplayLogp.apply(model_output, x, prev, k, bool(sde_solver))
    v = _as(model_output, _BF16)
    x0 = torch.empty_like(x)
    logp = torch.empty(B, dtype=_F32, device=dev)
    ws = ptr(logp_workspace(B, n, dev))
    if not grpo:
        mean = torch.empty_like(x)
        _dance_mean(x, v, mean, x0, logp, ws, B, n, k, sde_solver)
        return mean, x0
    if prev_sample is not None:
        prev = _as(prev_sample, _F32)
        check(lib().mgx_dance_step_fwd(ptr(x), ptr(v), None, ptr(prev), None, ptr(x0), ptr(logp), ws, B, n,
                                       C.byref(k), int(bool(sde_solver)), stream()))
        return prev, x0, logp
    out = prev_out if prev_out is not None else torch.empty_like(x)
    nz = None
    if sde_solver:
        nz = _as(noise, _F32) if noise is not None else torch.randn(x.shape, device=dev, dtype=_F32)
    check(lib().mgx_dance_step_fwd(ptr(x), ptr(v), ptr(nz), None, ptr(out), ptr(x0), ptr(logp), ws, B, n, C.byref(k),
                                   int(bool(sde_solver)), stream()))
    return out, x0, logp


def _dance_mean(x, v, mean, x0, logp, ws, B, n, k, sde_solver):
    """prev_sample_mean only: run the SDE kernel with zero noise scale so prev == mean."""
    kz = DanceCoeffs.from_buffer_copy(k)
    kz.sd = 0.0
    zeros = torch.zeros_like(x) if sde_solver else None
    check(lib().mgx_dance_step_fwd(ptr(x), ptr(v), ptr(zeros), None, ptr(mean), ptr(x0), ptr(logp), ws, B, n,
                                   C.byref(kz), int(bool(sde_solver)), stream()))


@dataclass
class DPMState:
    """History of x0 predictions for the multistep solver (reference sampling_utils.py:255-271)."""
    order: int
    model_outputs: List[Optional[torch.Tensor]] = None
    lower_order_nums: int = 0

    def __post_init__(self):
        self.model_outputs = [None] * self.order

    def update(self, model_output):
        self.model_outputs = self.model_outputs[1:] + [model_output]

    def update_lower_order(self):
        if self.lower_order_nums < self.order:
            self.lower_order_nums += 1


def convert_model_output(model_output, sample, sigmas, step_index):
    """x0 = sample - sigma*v (reference sampling_utils.py:387-396)."""
    x = _as(sample, _F32)
    v = _as(model_output, _BF16)
    out = torch.empty_like(x)
    check(lib().mgx_x0_pred(ptr(x), ptr(v), ptr(out), x.numel(), _r(_host(sigmas).to(_F32)[int(step_index)]), stream()))
    return out


def dpm_step(args, model_output, sample, step_index, timesteps, sigmas, dpm_state=None, generator=None,
             variance_noise=None, sde_solver=False, x_out=None):
    """DPM-Solver / DPM-Solver++ multistep step (reference sampling_utils.py:273-385) -> (prev, x0, log_prob)."""
    step_index = int(step_index)
    n_ts = len(timesteps)
    final = step_index == n_ts - 1
    second = step_index == n_ts - 2 and n_ts < 15
    dev = model_output.device
    x = _as(sample, _F32)
    v = _as(model_output, _BF16)
    B, n = _flat(x)
    if dpm_state:
        # the order decision is taken with this step's x0 already pushed (reference :313-357)
        lo = dpm_state.lower_order_nums
        if args.dpm_solver_order == 1 or lo < 1 or final:
            order = 1
        elif args.dpm_solver_order == 2 or lo < 2 or second:
            order = 2
        else:
            order = 3
        hist = dpm_state.model_outputs
        m1 = hist[-1] if order >= 2 else None     # before the push: [-1] is the previous step's x0
        m2 = hist[-2] if order >= 3 else None
    else:
        order, m1, m2 = 1, None, None
    noise = None
    if sde_solver:
        noise = (torch.randn(x.shape, generator=generator, device=dev, dtype=_F32) if variance_noise is None
                 else _as(variance_noise.to(dev), _F32))
    k = dpm_coeffs(args.dpm_algorithm_type, args.dpm_solver_type, order, sigmas, step_index, sde_solver)
    out = x_out if x_out is not None else torch.empty_like(x)
    if sde_solver and not dpm_state and model_output.requires_grad and torch.is_grad_enabled():
        # training replay under dpm_apply_strategy="all": the log-prob carries a gradient to the model output
        return _DpmSdeLogp.apply(model_output, x, noise, k, _u(_host(sigmas).to(_F32)[step_index]), out)
    x0 = torch.empty_like(x)
    logp = torch.empty(B, dtype=_F32, device=dev)
    check(lib().mgx_dpm_step_fwd(ptr(x), ptr(v), ptr(m1), ptr(m2), ptr(noise), ptr(out), ptr(x0), ptr(logp),
                                 ptr(logp_workspace(B, n, dev)), B, n, C.byref(k), stream()))
    if dpm_state is not None:
        dpm_state.update(x0)
        dpm_state.update_lower_order()
    return out, x0, logp


def flash_schedule(sigma_schedule, determistic, ratio, shift):
    """MixGRPO-Flash post-window compression (reference sampling_utils.py:33-54) on the host schedule."""
    sg = _host(sigma_schedule)
    n = sg.size(0)
    sde_idx = [i for i, d in enumerate(determistic) if not d]
    last = sde_idx[-1]
    num_post = int(max((n - 1 - last) * ratio, 1))
    t0 = torch.linspace(1, 0, n)[last + 1].item()
    post = sd3_time_shift(shift, torch.linspace(t0, 0, num_post))
    return torch.cat([sg[:last + 1], post], dim=0), last


def run_sample_step(args, z, progress_bar, sigma_schedule, transformer, encoder_hidden_states, pooled_prompt_embeds,
                    text_ids, image_ids, grpo_sample, determistic, noises=None, shared_rows=False):
    """T-step mixed ODE/SDE rollout (reference sampling_utils.py:12-155).

    Returns (z, latents, all_latents [B,T'+1,N,C] fp32, all_log_probs [B,T'] fp32).  `all_latents` is a
    transposed view of a step-major buffer [T'+1,B,N,C] (each solver step writes its output in place, and the
    training replay reads whole steps contiguously).  `noises` (optional) injects pre-drawn noise for parity
    tests, consumed in the reference's RNG order.

    `shared_rows=True` declares that all batch rows start identical (same x_T, same prompt: the reference's
    `init_same_noise` group).  Until the first step that injects per-sample noise the rows stay bit-identical, so
    the model and the solver run ONCE (batch 1) for those steps and the result is broadcast -- the reference
    recomputes the same values G times.  Outputs are unchanged.
    """
    dev = z.device
    noises = iter(noises) if noises is not None else None
    sig = _host(sigma_schedule).to(_F32)
    use_dpm = "dpmsolver" in args.dpm_algorithm_type
    post = use_dpm and args.dpm_apply_strategy == "post"
    state = DPMState(order=args.dpm_solver_order) if use_dpm else None
    last_sde = None
    if post:
        assert args.sample_strategy == "progressive", "post strategy is only supported for progressive sampling"
        sig, last_sde = flash_schedule(sig, determistic, args.dpm_post_compress_ratio, args.shift)
        progress_bar = range(sig.size(0) - 1)
    steps = sig.size(0) - 1
    B = encoder_hidden_states.shape[0]
    buf = torch.empty((steps + 1,) + tuple(z.shape), dtype=_F32, device=dev)
    buf[0].copy_(z)
    logps = torch.empty(steps, z.shape[0], dtype=_F32, device=dev)
    guidance = torch.tensor([3.5], device=dev, dtype=_BF16)
    txt_ids = text_ids.repeat(encoder_hidden_states.shape[1], 1)
    x0 = None
    n_run = 0
    # index of the first step whose update depends on per-sample noise (rows diverge AFTER that step's forward)
    first_noisy = steps
    if shared_rows and B > 1:
        for j in range(steps):
            in_flow = (not use_dpm) or (post and j <= last_sde)
            if (in_flow or not post) and not determistic[j]:
                first_noisy = j
                break
    else:
        first_noisy = -1
    ehs_all, pooled_all = encoder_hidden_states, pooled_prompt_embeds
    for i in progress_bar:
        one = shared_rows and B > 1 and i <= first_noisy       # forward at batch 1
        solve_one = shared_rows and B > 1 and i < first_noisy    # solver step at batch 1 too
        encoder_hidden_states = ehs_all[:1] if one else ehs_all
        pooled_prompt_embeds = pooled_all[:1] if one else pooled_all
        timestep_value = int(sig[i] * 1000)
        # int(sigma*1000)/1000 formed on the host as an IEEE fp32 division (reference :64-71 divides a long
        # tensor by 1000 on the device, where eager mode multiplies by the reciprocal instead)
        timestep = torch.full([1 if one else B], float(np.float32(timestep_value) / np.float32(1000)), device=dev,
                              dtype=_F32)
        transformer.eval()
        with torch.autocast("cuda", torch.bfloat16):
            pred = transformer(hidden_states=z[:1] if one else z, encoder_hidden_states=encoder_hidden_states,
                               timestep=timestep, guidance=guidance, txt_ids=txt_ids,
                               pooled_projections=pooled_prompt_embeds, img_ids=image_ids, joint_attention_kwargs=None,
                               return_dict=False)[0]
        if one and not solve_one:
            pred = pred.expand(B, -1, -1).contiguous()          # rows diverge in this step's update
            if state is not None:
                state.model_outputs = [None if m is None else m.expand(B, -1, -1).contiguous()
                                       for m in state.model_outputs]
        zf = buf[i]
        nxt = buf[i + 1]
        if solve_one:
            zf, nxt = zf[:1], nxt[:1]
        keep_x0 = args.drop_last_sample and i == steps - 1
        if (not use_dpm) or (post and i <= last_sde):
            if args.flow_grpo_sampling:
                nz = next(noises) if noises is not None else None
                if nz is not None and solve_one:
                    nz = nz[:1]
                _, x0_i, lp, _, _ = flow_grpo_step(pred, zf, args.eta, sig, i, None, determistic=determistic[i],
                                                   noise=nz, prev_out=nxt, want_x0=keep_x0 or post, want_mean=False)
                if post:  # feed the multistep history inside/before the window (reference :116-127)
                    state.update(x0_i)
                    state.update_lower_order()
            else:
                sde = not determistic[i]
                nz = next(noises) if (noises is not None and sde) else None
                _, x0_i, lp = dance_grpo_step(pred, zf, args.eta, sig, i, None, True, sde, noise=nz, prev_out=nxt)
        elif post:
            _, x0_i, lp = dpm_step(args, pred, zf, i, sig[:-1], sig, dpm_state=state, sde_solver=False, x_out=nxt)
        else:
            sde = not determistic[i]
            nz = next(noises) if (noises is not None and sde) else None
            _, x0_i, lp = dpm_step(args, pred, zf, i, sig[:-1], sig, dpm_state=state,
                                   generator=None if nz is not None else torch.Generator(device=dev),
                                   variance_noise=nz, sde_solver=sde, x_out=nxt)
        if solve_one:                                            # broadcast the shared result to all rows
            buf[i + 1][1:].copy_(nxt.expand(B - 1, -1, -1))
            nxt = buf[i + 1]
            lp = lp.expand(B)
            if x0_i is not None:
                x0_i = x0_i.expand(B, -1, -1)
        x0 = x0_i
        z = nxt
        logps[i].copy_(lp)
        n_run += 1
    latents = x0 if args.drop_last_sample else z
    return z, latents, buf.transpose(0, 1), logps.transpose(0, 1)
