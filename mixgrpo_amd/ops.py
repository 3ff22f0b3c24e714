"""Thin Python wrappers over the C-ABI MMDiT kernels (device tensors in, device tensors out, current stream).
Only plumbing lives here: shape bookkeeping and pointer passing.  See include/mixgrpo_hip.h for semantics."""
from dataclasses import dataclass

import torch

from ._lib import check, lib, ptr, stream

EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_GATE_RES, EPI_F32_ACC, EPI_DGELU = 0, 1, 2, 3, 4
BF16 = torch.bfloat16
F32 = torch.float32


@dataclass
class Rows:
    """A row-batched bf16/fp32 matrix view: row m at data_ptr + (m // rpb) * bstride + (m % rpb) * ld elements."""
    t: torch.Tensor      # tensor whose data_ptr() is the first element
    M: int               # number of rows
    ld: int
    rpb: int = 1 << 40
    bstride: int = 0

    @staticmethod
    def of(t):
        """Plain 2-D view of a contiguous tensor [..., cols]."""
        cols = t.shape[-1]
        return Rows(t, t.numel() // cols, cols)


def gemm(A: Rows, W, bias, C: Rows, N, K, epi=EPI_BIAS, gate=None, gate_ld=0, aux=None, beta=0.0, ldw=None):
    """C = epi(A @ W[N,K]^T + bias).  `aux` shares C's row map."""
    assert A.M == C.M
    check(lib().mgx_gemm_bf16(ptr(A.t), ptr(W), ptr(bias), ptr(C.t), ptr(gate), ptr(aux), A.M, N, K, A.ld, A.rpb,
                              A.bstride, K if ldw is None else ldw, C.ld, C.rpb, C.bstride, gate_ld, epi, beta, stream()))


_scratch = {}


def scratch(name, numel, dtype, device):
    """Grow-only named scratch buffers (caller-owned workspace for the C ABI)."""
    key = (name, dtype, device)
    t = _scratch.get(key)
    if t is None or t.numel() < numel:
        t = torch.empty(int(numel), dtype=dtype, device=device)
        _scratch[key] = t
    return t[:numel]


def transpose(inp: Rows, N, out, ld_out, colsum_out=None, colsum_beta=1.0):
    """out[N, ld_out] = inp[M, N]^T (zero padded); optional colsum_out[n] = beta*colsum_out[n] + sum_m inp[m,n]."""
    part = None
    if colsum_out is not None:
        part = scratch("colsum_partial", lib().mgx_transpose_partial_elems(inp.M, N), F32, out.device)
    check(lib().mgx_transpose_bf16(ptr(inp.t), ptr(out), ptr(part), ptr(colsum_out), colsum_beta, inp.M, N, inp.ld,
                                   inp.rpb, inp.bstride, ld_out, stream()))
