"""Data-parallel plumbing over torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in CPU tests).

The reference shards whole prompt groups across ranks (DistributedSampler, fastvideo/train_grpo_flux.py:737-739),
all-gathers rewards (:332-338) and relies on FSDP for gradient reduction.  Here every rank holds a full replica:
rewards are all-gathered the same way and gradients are summed with bucketed all-reduces over the flat fp32
gradient buffer (no per-parameter traffic, no parameter all-gathers).
"""
import os

import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if is_dist() else 1


def rank():
    return dist.get_rank() if is_dist() else 0


def main_print(*a, **k):
    """Rank-0 print gate (reference fastvideo/utils/logging_.py:8-10)."""
    if int(os.environ.get("LOCAL_RANK", "0")) <= 0:
        print(*a, **k)


def gather_tensor(t):
    """all_gather + cat along dim 0 (reference train_grpo_flux.py:332-338)."""
    if not is_dist():
        return t
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t.contiguous())
    return torch.cat(out, dim=0)


def allreduce_sum_(flat, bucket_elems=256 * 1024 * 1024, async_op=False):
    """In-place SUM all-reduce of a flat buffer in large buckets (xGMI rings are per-link bound: few, big messages).
    Returns the list of work handles when async_op."""
    if not is_dist() or dist.get_world_size() == 1:
        return []
    works = []
    n = flat.numel()
    for off in range(0, n, bucket_elems):
        w = dist.all_reduce(flat[off:min(n, off + bucket_elems)], op=dist.ReduceOp.SUM, async_op=async_op)
        if async_op:
            works.append(w)
    return works


def allreduce_mean_vec_(vec):
    """Average a small logging vector across ranks (one collective instead of the reference's 4 per inner step)."""
    if is_dist() and dist.get_world_size() > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        vec.div_(dist.get_world_size())
    return vec
