"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product (`mixgrpo_amd/`).

Plain PyTorch-CPU restatement of the reference's per-step solvers, written so that every
rounding point is explicit.  Pinned bit-exactly (fp32 outputs) against golden vectors produced
by the reference's own functions (tests/golden/gen_fixtures.py -> tests/golden/solver_steps.*).

Follows (file:line relative to /root/reference):
  sd3_time_shift            fastvideo/utils/sampling_utils.py:9-10
  flow_grpo_step            fastvideo/utils/sampling_utils.py:157-210
  dance_grpo_step           fastvideo/utils/sampling_utils.py:212-253
  DPMState / dpm_step       fastvideo/utils/sampling_utils.py:255-385
  convert_model_output      fastvideo/utils/sampling_utils.py:387-396
  dpm first/second/third    fastvideo/utils/sampling_utils.py:398-639

Dtype rules that matter, measured on torch-CPU where the fixtures were made:
  * a 0-dim fp32 tensor times a bf16 tensor is a bf16 tensor (product formed in fp32, rounded once);
  * written `scalar * tensor` the 0-dim operand is FIRST rounded to bf16 (it is cast to the common
    dtype like any tensor operand); written `tensor * scalar` it is used at full fp32 precision
    (TensorIterator's cpu-scalar fast path only covers the second operand).
`_scale` spells both out.  (On a CUDA/HIP device a 0-dim *device* tensor is never a "cpu scalar", so
there both spellings round the scalar to bf16; the product exposes that as `scalar_rounding="device"`,
see DESIGN.md.  The golden vectors pin the CPU behaviour.)
"""
import math
from dataclasses import dataclass, field
from typing import List, Optional

import torch

F32 = torch.float32
BF16 = torch.bfloat16


def sd3_time_shift(shift, t):
    return (shift * t) / (1 + (shift - 1) * t)


class _ScalarFirstMulBF16(torch.autograd.Function):
    """`c * t` (t bf16, c 0-dim fp32): forward rounds c to bf16 first; autograd's `grad * c` does not."""

    @staticmethod
    def forward(ctx, t, c):
        ctx.c = c
        return (t.to(F32) * c.to(BF16).to(F32)).to(BF16)

    @staticmethod
    def backward(ctx, g):
        return (g.to(F32) * ctx.c).to(g.dtype), None


def _scale(t: torch.Tensor, c, scalar_first: bool = True) -> torch.Tensor:
    """0-dim fp32 `c` times tensor `t`; `scalar_first` says how the reference spelled the product."""
    if t.dtype == BF16:
        if scalar_first:
            return _ScalarFirstMulBF16.apply(t, c if torch.is_tensor(c) else torch.tensor(c, dtype=F32))
        return (t.to(F32) * c).to(BF16)
    return t * c


def _gauss_logp(sample, mean, sd):
    """mean over non-batch dims of log N(sample; mean, sd^2), term order as sampling_utils.py:201-208."""
    lp = -((sample - mean) ** 2) / (2 * (sd ** 2)) - torch.log(sd) - torch.log(torch.sqrt(2 * torch.as_tensor(math.pi)))
    return lp.mean(dim=tuple(range(1, lp.ndim)))


@dataclass
class FlowCoeffs:
    sigma: torch.Tensor
    dt: torch.Tensor
    std: torch.Tensor       # eta*sqrt(sigma/(1-sigma)), sigma==1 -> uses sigmas[1] in the denominator
    c_x: torch.Tensor       # multiplies latents in the SDE mean
    c_v: torch.Tensor       # multiplies model_output (before *dt)
    sd: torch.Tensor        # std*sqrt(-dt): the Gaussian's standard deviation


def flow_coeffs(sigmas: torch.Tensor, index: int, eta: float) -> FlowCoeffs:
    s = sigmas[index]
    dt = sigmas[index + 1] - s
    s_den = torch.where(s == 1, sigmas[1], s)
    std = torch.sqrt(s / (1 - s_den)) * eta
    return FlowCoeffs(s, dt, std, 1 + std ** 2 / (2 * s) * dt, 1 + std ** 2 * (1 - s) / (2 * s),
                      std * torch.sqrt(-1 * dt))


def flow_grpo_step(model_output, latents, eta, sigmas, index, prev_sample, generator=None, determistic=False,
                   noise=None):
    """Returns (prev_sample, x0, log_prob[B], mean, sd).  `noise` lets tests inject the draw."""
    if prev_sample is not None and generator is not None:
        raise ValueError("Cannot pass both generator and prev_sample.")
    k = flow_coeffs(sigmas, index, eta)
    x0 = latents - _scale(model_output, k.sigma)
    mean = latents * k.c_x + _scale(_scale(model_output, k.c_v, False), k.dt, False)
    if prev_sample is None:
        if noise is None:
            noise = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype)
        prev_sample = mean + _scale(noise, k.sd)
    if determistic:
        prev_sample = latents + _scale(model_output, k.dt)
    return prev_sample, x0, _gauss_logp(prev_sample.detach(), mean, k.sd), mean, k.sd


def dance_grpo_step(model_output, latents, eta, sigmas, index, prev_sample, grpo, sde_solver, noise=None):
    s = sigmas[index]
    ds = sigmas[index + 1] - s
    mean = latents + _scale(model_output, ds)
    x0 = latents - _scale(model_output, s)
    sd = eta * math.sqrt(s - sigmas[index + 1])          # python float
    if sde_solver:
        score = -(latents - x0 * (1 - s)) / s ** 2
        mean = mean + (-0.5 * eta ** 2 * score) * ds
    if grpo and prev_sample is None:
        if sde_solver:
            if noise is None:
                noise = torch.randn_like(mean)
            prev_sample = mean + noise * sd
        else:
            prev_sample = mean
    if not grpo:
        return mean, x0
    # the reference's normaliser terms are a dead expression (sampling_utils.py:247): only the quadratic survives
    lp = -((prev_sample.detach().to(F32) - mean.to(F32)) ** 2) / (2 * (sd ** 2))
    return prev_sample, x0, lp.mean(dim=tuple(range(1, lp.ndim)))


# ----------------------------------------------------------------------------- DPM-Solver(++)
@dataclass
class DPMState:
    order: int
    model_outputs: List[Optional[torch.Tensor]] = None
    lower_order_nums: int = 0

    def __post_init__(self):
        self.model_outputs = [None] * self.order

    def update(self, x0):
        self.model_outputs = self.model_outputs[1:] + [x0]

    def update_lower_order(self):
        self.lower_order_nums = min(self.lower_order_nums + 1, self.order)


def convert_model_output(model_output, sample, sigmas, step_index):
    return sample - _scale(model_output, sigmas[step_index])


def _lam(sig):
    return torch.log(1 - sig) - torch.log(sig)


def _dpm_update(algo, stype, order, hist, sigmas, i, sample, noise, sde):
    """One multistep update of the requested order; returns (x_t, mean, std, dt_sqrt)."""
    sig_t, sig_s0 = sigmas[i + 1], sigmas[i]
    a_t, a_s0 = 1 - sig_t, 1 - sig_s0
    lam_t, lam_s0 = _lam(sig_t), _lam(sig_s0)
    h = lam_t - lam_s0
    D0 = hist[-1]
    D1 = D2 = None
    if order >= 2:
        lam_s1 = _lam(sigmas[i - 1])
        h0 = lam_s0 - lam_s1
        r0 = h0 / h
        D1_0 = (1.0 / r0) * (hist[-1] - hist[-2])
        D1 = D1_0
    if order == 3:
        lam_s2 = _lam(sigmas[i - 2])
        r1 = (lam_s1 - lam_s2) / h
        D1_1 = (1.0 / r1) * (hist[-2] - hist[-3])
        D1 = D1_0 + (r0 / (r0 + r1)) * (D1_0 - D1_1)
        D2 = (1.0 / (r0 + r1)) * (D1_0 - D1_1)
    std = sig_t
    if algo == "dpmsolver++":
        e2 = 1 - torch.exp(-2.0 * h)
        mean = (sig_t / sig_s0 * torch.exp(-h)) * sample + (a_t * e2) * D0
        if order == 2 and stype == "midpoint":
            mean = mean + 0.5 * (a_t * e2) * D1
        elif order >= 2:
            mean = mean + (a_t * ((1.0 - torch.exp(-2.0 * h)) / (-2.0 * h) + 1.0)) * D1
        if order == 3:
            mean = mean + (a_t * ((1.0 - torch.exp(-2.0 * h) - 2.0 * h) / (2.0 * h) ** 2 - 0.5)) * D2
        dt_sqrt = torch.sqrt(1.0 - torch.exp(-2 * h))
        if sde:
            assert noise is not None
            return mean + std * dt_sqrt * noise, mean, std, dt_sqrt
        em1 = torch.exp(-h) - 1.0
        x = (sig_t / sig_s0) * sample - (a_t * em1) * D0
        if order == 2 and stype == "midpoint":
            x = x - 0.5 * (a_t * em1) * D1
        elif order >= 2:
            x = x + (a_t * (em1 / h + 1.0)) * D1
        if order == 3:
            x = x - (a_t * ((torch.exp(-h) - 1.0 + h) / h ** 2 - 0.5)) * D2
        return x, mean, std, dt_sqrt
    if algo == "dpmsolver":
        if order == 3:
            raise NotImplementedError("reference third-order 'dpmsolver' is unreachable (sampling_utils.py:629-639)")
        eh1 = torch.exp(h) - 1.0
        mean = (a_t / a_s0) * sample - 2.0 * (sig_t * eh1) * D0
        if order == 2 and stype == "midpoint":
            mean = mean - (sig_t * eh1) * D1
        elif order == 2:
            mean = mean - 2.0 * (sig_t * (eh1 / h - 1.0)) * D1
        dt_sqrt = torch.sqrt(torch.exp(2 * h) - 1.0)
        if sde:
            assert noise is not None
            return mean + std * dt_sqrt * noise, mean, std, dt_sqrt
        x = (a_t / a_s0) * sample - (sig_t * eh1) * D0
        if order == 2 and stype == "midpoint":
            x = x - 0.5 * (sig_t * eh1) * D1
        elif order == 2:
            x = x - (sig_t * (eh1 / h - 1.0)) * D1
        return x, mean, std, dt_sqrt
    raise ValueError(algo)


def dpm_step(args, model_output, sample, step_index, timesteps, sigmas, dpm_state=None, generator=None,
             variance_noise=None, sde_solver=False):
    n = len(timesteps)
    final = step_index == n - 1
    second = step_index == n - 2 and n < 15
    x0 = convert_model_output(model_output, sample, sigmas, step_index)
    if dpm_state is not None:
        dpm_state.update(x0)
    sample = sample.to(F32)
    noise = None
    if sde_solver:
        noise = (torch.randn(x0.shape, generator=generator, dtype=F32) if variance_noise is None
                 else variance_noise.to(F32))
    if dpm_state:
        if args.dpm_solver_order == 1 or dpm_state.lower_order_nums < 1 or final:
            order = 1
        elif args.dpm_solver_order == 2 or dpm_state.lower_order_nums < 2 or second:
            order = 2
        else:
            order = 3
        hist = dpm_state.model_outputs
    else:
        order, hist = 1, [x0]
    x_t, mean, std, dt_sqrt = _dpm_update(args.dpm_algorithm_type, args.dpm_solver_type, order, hist, sigmas,
                                          step_index, sample, noise, sde_solver)
    if dpm_state is not None:
        dpm_state.update_lower_order()
    x_t = x_t.to(x0.dtype)
    return x_t, x0, _gauss_logp(x_t.detach(), mean, std * dt_sqrt)
