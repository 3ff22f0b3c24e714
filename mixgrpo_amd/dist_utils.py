"""Data-parallel plumbing over torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in CPU tests).

The reference shards whole prompt groups across ranks (DistributedSampler, fastvideo/train_grpo_flux.py:737-739),
all-gathers rewards (:332-338) and relies on FSDP for gradient reduction.  Here every rank holds a full replica:
rewards are all-gathered the same way and gradients are summed with bucketed all-reduces over the flat fp32
gradient buffer (no per-parameter traffic, no parameter all-gathers).

Gradient reduction (`GradReducer`): the flat fp32 gradient buffer is summed over the ranks in BUCKETS that follow the
parameter store's block order.  Modes:
  "fp32"  (default) the buffer itself is all-reduced in place, like the reference's FSDP, which reduce-scatters its gradients
          in fp32 (fastvideo/utils/fsdp_util.py:56-66): 47.6 GB per optimizer step for FLUX.1-dev
  "bf16"  (MGX_DP_GRAD_DTYPE=bf16 / transformer.dp_grad_dtype) each bucket is cast to bf16 (one HIP pass), all-reduced, and
          cast back into the fp32 buffer: half the xGMI bytes; the sum of `world` bf16 values carries a relative error of
          ~2^-9 per element (two-rank test: tests/test_hip_dp.py; never measured at 8 ranks: opt-in)
With `overlap=True` the buckets of a transformer block are launched (async, RCCL's own stream) as soon as the LAST
micro-batch's backward has finished that block -- the reference's FSDP reduce-scatters per wrapped block during the
backward too (fastvideo/utils/fsdp_util.py:56-66) -- and only waited for before the optimizer step.  Overlap is opt-in:
RCCL's channels take CUs away from the persistent GEMM (one workgroup per CU), which has not been measurable here (no
multi-GPU node behind gpurun); see DESIGN.md section 5.
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


class GradReducer:
    """SUM all-reduce of a flat fp32 gradient buffer in buckets; see the module docstring for the modes."""

    def __init__(self, flat, mode=None, bucket_elems=128 * 1024 * 1024, overlap=None):
        self.flat = flat
        # default fp32: the reference's FSDP reduces its gradients in fp32 (fastvideo/utils/fsdp_util.py:56-66), and a ring over
        # 8 ranks would sum seven bf16 partials.  bf16 (half the xGMI bytes) is an opt-in until an 8-GPU measurement says the
        # bytes matter (even the fp32 ring is < 5 % of a step).
        self.mode = mode or os.environ.get("MGX_DP_GRAD_DTYPE", "fp32")
        if self.mode not in ("bf16", "fp32"):
            raise ValueError(f"gradient all-reduce mode {self.mode!r} (expected 'bf16' or 'fp32')")
        self.overlap = bool(int(os.environ.get("MGX_DP_OVERLAP", "0"))) if overlap is None else bool(overlap)
        self.bucket_elems = int(bucket_elems) // 64 * 64
        self.max_in_flight = 8       # buckets (x 2 B x bucket_elems of staging each: 2 GiB at the defaults)
        self.pending = []            # (work handle, lo, hi, staging buffer) of buckets in flight
        self._staging = []           # reusable bf16 staging buffers
        self.launched = []           # ranges already launched in this optimizer step (overlap mode)

    # -- one bucket ------------------------------------------------------------------------------------------------
    def _launch(self, lo, hi, async_op):
        from . import ops
        g = self.flat[lo:hi]
        if self.mode == "fp32":
            w = dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=async_op)
            if async_op:
                self.pending.append((w, lo, hi, None))
            return
        while len(self.pending) >= self.max_in_flight:           # bound the staging memory: retire the oldest bucket first
            self._retire(self.pending.pop(0))
        buf = self._staging.pop() if self._staging else torch.empty(self.bucket_elems, dtype=torch.bfloat16, device=g.device)
        b = buf[:hi - lo]
        ops.cast_bf16(g, b)
        w = dist.all_reduce(b, op=dist.ReduceOp.SUM, async_op=async_op)
        if async_op:
            self.pending.append((w, lo, hi, buf))
        else:
            ops.cast_f32(b, g)
            self._staging.append(buf)

    def reduce_range(self, lo, hi, async_op=False):
        """Sum flat[lo:hi] over the ranks (bucketed).  lo / hi are multiples of 64 (the store's alignment)."""
        if not is_dist() or dist.get_world_size() == 1 or hi <= lo:
            return
        for off in range(lo, hi, self.bucket_elems):
            self._launch(off, min(hi, off + self.bucket_elems), async_op)
        self.launched.append((lo, hi))

    def _retire(self, item):
        from . import ops
        w, lo, hi, buf = item
        w.wait()                                     # the current stream waits for the collective; no host block
        if buf is not None:
            ops.cast_f32(buf[:hi - lo], self.flat[lo:hi])
            self._staging.append(buf)

    def finish(self):
        """Reduce whatever part of the buffer has not been launched yet (everything, without overlap), wait for the buckets
        in flight and convert them back.  After this the whole buffer holds the sum over the ranks.  Every rank issues the
        same collectives in the same order (launch order follows the backward pass, which is identical on all ranks)."""
        n = self.flat.numel()
        launched, self.launched = sorted(self.launched), []
        gaps, pos = [], 0
        for lo, hi in launched:
            if lo > pos:
                gaps.append((pos, lo))
            pos = max(pos, hi)
        if pos < n:
            gaps.append((pos, n))
        for lo, hi in gaps:
            self.reduce_range(lo, hi, async_op=False)
        self.launched = []
        for item in self.pending:
            self._retire(item)
        self.pending = []
