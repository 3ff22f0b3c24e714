"""world_size-2 gloo tests (CPU) of the data-parallel plumbing: reward all-gather, bucketed gradient all-reduce,
logging-vector averaging, per-rank prompt partitioning and the DP-equivalence of averaged gradients."""
import os

import pytest
import torch
import torch.distributed as dist

from helpers import run_ranks


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mixgrpo_amd import dist_utils as DU
    out = {}
    # rewards: each rank owns one whole group of G=4 (reference partitioning) -> gathered [world*G]
    r = torch.arange(4, dtype=torch.float32) + 10 * rank
    out["gathered"] = DU.gather_tensor(r).tolist()
    # flat gradient buffer reduced in several buckets (sizes that do not divide evenly)
    g = torch.full((1000,), float(rank + 1))
    g[::7] += rank
    DU.allreduce_sum_(g, bucket_elems=300)
    out["gsum_head"] = g[:8].tolist()
    works = DU.allreduce_sum_(torch.ones(10), async_op=True)
    for w in works:
        w.wait()
    # GradReducer (fp32 mode runs on host tensors too): two block ranges launched asynchronously "during the backward", the
    # gaps (head of the buffer, a middle range, the tail) picked up by finish(); every element summed exactly once
    flat = torch.arange(64 * 40, dtype=torch.float32) * (rank + 1)
    red = DU.GradReducer(flat, mode="fp32", bucket_elems=64 * 3, overlap=True)
    red.reduce_range(64 * 30, 64 * 36, async_op=True)
    red.reduce_range(64 * 10, 64 * 20, async_op=True)
    red.finish()
    out["reducer_ok"] = bool(torch.equal(flat, torch.arange(64 * 40, dtype=torch.float32) * 3)) and not red.pending
    red.finish()                                     # nothing launched: reduces the whole buffer once more
    out["reducer_again"] = bool(torch.equal(flat, torch.arange(64 * 40, dtype=torch.float32) * 6))
    v = torch.tensor([1.0, 2.0, 3.0, 4.0]) * (rank + 1)
    out["mean_vec"] = DU.allreduce_mean_vec_(v).tolist()
    out["rank_world"] = (DU.rank(), DU.world_size(), DU.is_dist())
    # DP equivalence: mean of per-rank grads == grad of the mean loss over the union of the two shards
    torch.manual_seed(0)
    w0 = torch.randn(5, requires_grad=True)
    x = torch.randn(2, 3, 5)[rank]
    (x @ w0).pow(2).mean().backward()
    gl = w0.grad.clone()
    dist.all_reduce(gl)
    gl /= world
    out["dp_grad"] = gl.tolist()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out))


def test_world_size_2_gloo():
    res = dict(run_ranks(_worker, 2, timeout=120))
    for rk in (0, 1):
        o = res[rk]
        assert o["gathered"] == [0.0, 1.0, 2.0, 3.0, 10.0, 11.0, 12.0, 13.0]
        assert o["gsum_head"] == [4.0, 3.0, 3.0, 3.0, 3.0, 3.0, 3.0, 4.0]
        assert o["mean_vec"] == [1.5, 3.0, 4.5, 6.0]
        assert o["rank_world"] == (rk, 2, True)
        assert o["reducer_ok"] and o["reducer_again"]
    # reference: full-batch gradient of the mean of the two per-shard losses
    torch.manual_seed(0)
    w0 = torch.randn(5, requires_grad=True)
    xs = torch.randn(2, 3, 5)
    (0.5 * ((xs[0] @ w0).pow(2).mean() + (xs[1] @ w0).pow(2).mean())).backward()
    assert torch.allclose(torch.tensor(res[0]["dp_grad"]), w0.grad, atol=1e-6)
    assert res[0]["dp_grad"] == res[1]["dp_grad"]


def test_single_process_fallbacks():
    from mixgrpo_amd import dist_utils as DU
    t = torch.arange(3.0)
    assert DU.gather_tensor(t) is t
    assert DU.allreduce_sum_(t.clone()) == []
    assert DU.world_size() == 1 and DU.rank() == 0
