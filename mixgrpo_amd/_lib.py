"""ctypes binding of libmixgrpo_hip.so (the C ABI declared in include/mixgrpo_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, this raises.
PyTorch is used only for device memory (tensor.data_ptr()) and the current HIP stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmixgrpo_hip.so")


class MgxError(RuntimeError):
    pass


class FlowCoeffs(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("sigma_x0", "c_x", "c_v", "dt_mean", "sd_noise", "dt_det", "den", "log_sd",
                                         "log_c")]


class DanceCoeffs(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("ds_r", "s_r", "ds", "ds_b", "s_b", "one_m_s", "s_sq", "half_eta2", "sd", "den")]


class DpmCoeffs(C.Structure):
    _fields_ = [("order", C.c_int), ("sde", C.c_int), ("sigma_x0", C.c_float), ("inv_r0", C.c_float),
                ("inv_r1", C.c_float), ("c_r", C.c_float), ("inv_r01", C.c_float), ("cm", C.c_float * 4),
                ("cx", C.c_float * 4), ("sd_noise", C.c_float), ("den", C.c_float), ("log_sd", C.c_float),
                ("log_c", C.c_float)]


_P = C.c_void_p
_I = C.c_int
_L = C.c_long
_F = C.c_float
_D = C.c_double

# name -> (restype, argtypes); mirrors include/mixgrpo_hip.h one to one
SIGNATURES = {
    "mgx_version": (_I, []),
    "mgx_last_error": (C.c_char_p, []),
    "mgx_logp_workspace_elems": (_L, [_I, _L]),
    "mgx_flow_step_fwd": (_I, [_P] * 9 + [_I, _L, C.POINTER(FlowCoeffs), _I, _P]),
    "mgx_flow_step_bwd": (_I, [_P] * 5 + [_I, _L, C.POINTER(FlowCoeffs), _P]),
    "mgx_dance_step_fwd": (_I, [_P] * 8 + [_I, _L, C.POINTER(DanceCoeffs), _I, _P]),
    "mgx_dance_step_bwd": (_I, [_P] * 5 + [_I, _L, C.POINTER(DanceCoeffs), _I, _P]),
    "mgx_dpm_step_fwd": (_I, [_P] * 9 + [_I, _L, C.POINTER(DpmCoeffs), _P]),
    "mgx_dpm_step_bwd": (_I, [_P] * 5 + [_I, _L, C.POINTER(DpmCoeffs), _F, _P]),
    "mgx_x0_pred": (_I, [_P, _P, _P, _L, _F, _P]),
    "mgx_pack_latents": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "mgx_unpack_latents": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "mgx_group_advantage": (_I, [_P, _P, _I, _I, _F, _F, _I, _P]),
    "mgx_global_advantage": (_I, [_P, _P, _P, _I, _I, _P]),
    "mgx_grpo_loss": (_I, [_P, _P, _P, _I, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P]),
    "mgx_gemm_bf16": (_I, [_P] * 6 + [_L, _I, _I, _I] + [_L] * 8 + [_I, _F, _P]),
    "mgx_gemm_sk_workspace_elems": (_L, []),
    "mgx_gemm_bf16_sk": (_I, [_P] * 6 + [_L, _I, _I, _I] + [_L] * 8 + [_I, _F, _P, _L, _P]),
    "mgx_linear_bf16_t": (_I, [_P] * 4 + [_I] * 3 + [_L] * 5 + [_P, _L, _P]),
    "mgx_linear_qk_norm_rope": (_I, [_P] * 10 + [_I] * 6 + [_L, _L, _F, _P]),
    "mgx_gemm_bf16_pair": (_I, ([_P] * 6 + [_I] + [_L] * 4) * 2 + [_I, _I] + [_L] * 5 + [_I, _F, _P, _L, _P]),
    "mgx_transpose_partial_elems": (_L, [_I, _I]),
    "mgx_transpose_bf16": (_I, [_P, _P, _P, _P, _F, _I, _I, _L, _L, _L, _L, _P]),
    "mgx_ln_modulate_fwd": (_I, [_P, _L, _L, _L, _P, _P, _L, _P, _L, _P, _L, _I, _P]),
    "mgx_ln_modulate_bwd_workspace": (_L, [_L, _L, _I]),
    "mgx_ln_modulate_bwd": (_I, [_P, _L, _P, _L, _L, _L, _P, _L, _P, _L, _L, _L, _I, _P, _P, _P, _L, _I, _P]),
    "mgx_qk_norm_rope_fwd": (_I, [_P, _L] + [_P] * 10 + [_I] * 6 + [_P]),
    "mgx_qk_norm_rope_fwd_qs": (_I, [_P, _L] + [_P] * 10 + [_I] * 6 + [_F, _P]),
    "mgx_qk_norm_rope_bwd_workspace": (_L, [_I, _I, _I]),
    "mgx_qk_norm_rope_bwd": (_I, [_P, _L] + [_P] * 8 + [_L] + [_P] * 3 + [_I] * 6 + [_P]),
    "mgx_qk_norm_rope_bwd_qs": (_I, [_P, _L] + [_P] * 8 + [_L] + [_P] * 3 + [_I] * 6 + [_F, _P]),
    "mgx_attn_fwd": (_I, [_P] * 5 + [_I] * 4 + [_L, _L, _F, _P]),
    "mgx_attn_fwd_log2": (_I, [_P] * 5 + [_I] * 4 + [_L, _L, _P]),
    "mgx_attn_fp8_quantize": (_I, [_P] * 7 + [_I] * 4 + [_P]),
    "mgx_attn_fwd_fp8": (_I, [_P] * 6 + [_I] * 4 + [_L, _L, _F, _P]),
    "mgx_attn_bwd": (_I, [_P] * 13 + [_I] * 4 + [_L, _L, _F, _P]),
    "mgx_skinny_linear": (_I, [_P, _L, _P, _L, _P, _P, _L, _I, _I, _I, _P]),
    "mgx_skinny_wgrad": (_I, [_P, _L, _P, _L, _P, _L, _P, _I, _I, _I, _P]),
    "mgx_skinny_dgrad_workspace": (_L, [_I, _I]),
    "mgx_conv3x3_nhwc": (_I, [_P, _P, _P, _P, _L, _P, _I, _I, _I, _I, _I, _P]),
    "mgx_group_norm_workspace": (_L, [_L, _I, _I]),
    "mgx_group_norm_nhwc": (_I, [_P, _L, _P, _P, _P, _L, _L, _P, _I, _I, _I, _I, _F, _I, _P]),
    "mgx_upsample2x_pad_nhwc": (_I, [_P, _P, _I, _I, _I, _P]),
    "mgx_latents_to_pad_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "mgx_nhwc_to_image": (_I, [_P, _L, _P, _I, _I, _I, _P]),
    "mgx_softmax_rows_f32": (_I, [_P, _L, _P, _L, _I, _I, _F, _P]),
    "mgx_skinny_dgrad": (_I, [_P, _L, _P, _L, _P, _L, _P, _I, _I, _I, _I, _P]),
    "mgx_ew_bf16": (_I, [_P, _P, _P, _L, _I, _P]),
    "mgx_sincos_embed": (_I, [_P, _P, _I, _P]),
    "mgx_cast_f32_bf16": (_I, [_P, _P, _L, _P]),
    "mgx_cast_bf16_f32": (_I, [_P, _P, _L, _F, _P]),
    "mgx_gelu_bf16": (_I, [_P, _L, _P, _L, _L, _I, _P]),
    "mgx_transpose_gelu_bf16": (_I, [_P, _P, _I, _I, _L, _L, _P]),
    "mgx_gate_bwd_workspace": (_L, [_L, _L, _I]),
    "mgx_gate_bwd": (_I, [_P, _L, _L, _P, _L, _P, _L, _P, _L, _P, _P, _I, _L, _I, _P]),
    "mgx_sqnorm_workspace": (_L, []),
    "mgx_sqnorm_f32": (_I, [_P, _L, _P, _P, _F, _P]),
    "mgx_adamw_step": (_I, [_P, _P, _P, _P, _P, _L, _D, _D, _D, _D, _D, _I, _P, _F, _F, _P]),
    "mgx_scale_f32": (_I, [_P, _L, _F, _P]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises MgxError when the HIP extension is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MgxError(f"{LIB_PATH} is missing: run `python -m mixgrpo_amd.build` (hipcc, gfx950). "
                           "There is no CPU fallback for the product path.")
        h = C.CDLL(LIB_PATH)
        h.mgx_version.restype = _I
        v = h.mgx_version()
        if v < 0:
            raise MgxError(f"{LIB_PATH} is a DIAGNOSTIC build (mgx_version() = {v}, -DMGX_DIAGNOSTIC_BUILD): its timing-only "
                           "switches compute wrong results.  Rebuild with `python -m mixgrpo_amd.build --force`.")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(code):
    if code != 0:
        raise MgxError(f"libmixgrpo_hip error {code}: {lib().mgx_last_error().decode()}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses CPU tensors: kernels only see device memory."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MgxError("mixgrpo_amd HIP ops need device tensors (got a CPU tensor); there is no CPU path")
    if not t.is_contiguous():
        raise MgxError("mixgrpo_amd HIP ops need contiguous tensors")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


_ws_cache = {}


def logp_workspace(B, n, device):
    """Reusable fp64 workspace for the log-prob reduction (caller-owned memory on the C ABI)."""
    need = lib().mgx_logp_workspace_elems(B, n)
    key = (device, torch.cuda.current_stream().cuda_stream)
    w = _ws_cache.get(key)
    if w is None or w.numel() < need:
        w = torch.empty(max(need, 4096), dtype=torch.float64, device=device)
        _ws_cache[key] = w
    return w
