"""Shared test helpers: fixture loading."""
import json
import os

from safetensors.torch import load_file

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    tensors = load_file(os.path.join(GOLDEN, name + ".safetensors"))
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        meta = json.load(f)
    return tensors, meta


def host_matches_fixture_host():
    """torch's CPU scalar math (sqrt/log of 0-dim tensors) is not bit-stable across hosts: the fixtures carry the
    generating host's last bits.  True when this host reproduces them (then comparisons are bit-exact)."""
    import torch
    t, _ = load_golden("solver_steps")
    sig = t["sigma/shift3.0_T25"]
    s, dt = sig[2], sig[3] - sig[2]
    std = torch.sqrt(s / (1 - s)) * 0.7
    return (1 + std ** 2 / (2 * s) * dt).item() == 0.8668485283851624 and \
        torch.log(std * torch.sqrt(-1 * dt)).item() == -0.6758469939231873


def assert_same(a, b, exact=True, rtol=2e-6, atol=2e-6):
    """Bit-exact comparison (NaNs equal), or a few-ulp tolerance when `exact` is False."""
    import torch
    a = a.detach().cpu()
    assert a.dtype == b.dtype and a.shape == b.shape, (a.dtype, b.dtype, a.shape, b.shape)
    assert torch.equal(torch.isnan(a), torch.isnan(b))
    if exact:
        assert torch.equal(torch.nan_to_num(a, nan=777.0), torch.nan_to_num(b, nan=777.0)), (a - b).abs().max()
    else:
        fin = torch.isfinite(b)
        assert torch.equal(a[~fin & ~torch.isnan(b)], b[~fin & ~torch.isnan(b)])
        assert torch.allclose(a[fin].float(), b[fin].float(), rtol=rtol, atol=atol), (a[fin].float() - b[fin].float()).abs().max()


def oracle_flux(cfg, P, device=None):
    """oracle/mmdit.forward behind the transformer call signature, parameters trainable (fp32 master weights; the
    restatement rounds to bf16 where autocast does).  Used by the end-to-end tests on both the CPU and the GPU side.
    `device`: run the MMDiT restatement's (device-agnostic, plain fp32 torch) arithmetic there -- parameters on that device,
    inputs moved over, the output moved back -- while the trainer oracle around it stays on the host: what makes the
    full-width, many-block cases affordable inside the suite's time limit (the host pass of a 2.5 B-parameter train step took
    100-340 s depending on the box).  None: everything on the host."""
    import torch
    from oracle import mmdit as OM

    class OracleFlux(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.cfg, self.names = cfg, list(P)
            self.params = torch.nn.ParameterList([torch.nn.Parameter(P[k].clone() if device is None else P[k].to(device))
                                                  for k in self.names])
            self.config = {"oracle": True}

        def forward(self, hidden_states, encoder_hidden_states, timestep, guidance, txt_ids, pooled_projections, img_ids,
                    joint_attention_kwargs=None, return_dict=False):
            Pd = dict(zip(self.names, self.params))
            f = (lambda x: x.float()) if device is None else (lambda x: x.float().to(device))
            out = OM.forward(Pd, self.cfg, f(hidden_states), f(encoder_hidden_states), f(timestep), f(guidance), f(txt_ids),
                             f(pooled_projections), f(img_ids))
            out = out.to(torch.bfloat16)
            return (out if device is None else out.cpu(),)

        def clip_grad_norm_(self, max_norm):
            n = torch.nn.utils.clip_grad_norm_(self.parameters(), max_norm)
            return n if device is None else n.cpu()

    return OracleFlux()


def free_port():
    """A TCP port the kernel just handed out (bind to port 0): distinct per call, whatever the pytest pid is."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(worker, world, timeout=300, extra=()):
    """Start `worker(rank, world, port, queue, *extra)` as `world` daemon processes, collect one queue item per rank and
    ALWAYS reap the children: a rank that fails can neither hang the collection (bounded `get`) nor pytest's exit (daemon
    processes, terminate() + join() in `finally`).  Returns the items sorted by their first field (the rank)."""
    import queue as _queue

    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q) + tuple(extra), daemon=True) for r in range(world)]
    try:
        for p in procs:
            p.start()
        out = []
        for _ in procs:
            try:
                out.append(q.get(timeout=timeout))
            except _queue.Empty:
                codes = [p.exitcode for p in procs]
                raise AssertionError(f"a rank produced no result within {timeout} s (exit codes so far: {codes})")
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0, f"rank exited with {p.exitcode}"
        return sorted(out, key=lambda t: t[0])
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(timeout=10)
            if p.is_alive():
                p.kill()
