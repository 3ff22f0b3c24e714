"""Do two streams of half-batch GEMM chains pack each other's partial last rounds?  One stream: the FLUX block's four Linears at
M = 36864 (rollout batch 8); two streams: the same chains at M = 18432 each, launched alternately, run concurrently.
Same FLOPs; the question is wall time.  Also the attention forward at B = 8 against 2 x B = 4."""
import json, math, sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import ops
from mixgrpo_amd.ops import Rows
torch.manual_seed(0)
dev = "cuda"
d = 3072
SH = [(3 * d, d, 0), (d, d, 2), (4 * d, d, 1), (d, 4 * d, 2)]        # (N, K, epi): QKV, to_out, ff1 (GELU), ff2 (gate-residual)


def make(M):
    bufs = []
    for (N, K, epi) in SH:
        A = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
        b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
        C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        gate = torch.ones(1, N, device=dev, dtype=torch.bfloat16) if epi == 2 else None
        bufs.append((A, W, b, C, gate, N, K, epi, M))
    return bufs


def chain(bufs):
    for (A, W, b, C, gate, N, K, epi, M) in bufs:
        ops.gemm(Rows.of(A), W, b, Rows.of(C), N, K, epi, gate=gate, gate_ld=N)


def wall(fn, n=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


full = make(36864)
h1, h2 = make(18432), make(18432)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
REP = 6


def one_stream():
    for _ in range(REP):
        chain(full)


def one_stream_halves():
    for _ in range(REP):
        chain(h1); chain(h2)


def two_streams():
    main = torch.cuda.current_stream()
    s1.wait_stream(main); s2.wait_stream(main)
    for _ in range(REP):
        with torch.cuda.stream(s1):
            chain(h1)
        with torch.cuda.stream(s2):
            chain(h2)
    main.wait_stream(s1); main.wait_stream(s2)


def two_streams_fine():
    """alternate launch by launch"""
    main = torch.cuda.current_stream()
    s1.wait_stream(main); s2.wait_stream(main)
    for _ in range(REP):
        for x1, x2 in zip(h1, h2):
            with torch.cuda.stream(s1):
                chain([x1])
            with torch.cuda.stream(s2):
                chain([x2])
    main.wait_stream(s1); main.wait_stream(s2)


fl = sum(2.0 * 36864 * N * K for (N, K, _) in SH) * REP / 1e9
res = {}
for rep in range(3):
    for name, fn in (("one_stream_M36864", one_stream), ("one_stream_2xM18432", one_stream_halves), ("two_streams_chain", two_streams),
                     ("two_streams_alternating", two_streams_fine)):
        res.setdefault(name, []).append(wall(fn, 3))
for k, v in res.items():
    print(json.dumps({"case": k, "ms": round(min(v), 3), "tflops": round(fl / min(v))}), flush=True)

# attention forward
C = 1.4426950408889634 / math.sqrt(128)
H, S = 24, 4608
def attn_bufs(B):
    q, k, v = (torch.randn(B, H, S, 128, device=dev).bfloat16() for _ in range(3))
    return ((q.float() * C).bfloat16(), k, v.transpose(-1, -2).contiguous(), torch.empty(B, S, H * 128, device=dev, dtype=torch.bfloat16), B)
a8, a4a, a4b = attn_bufs(8), attn_bufs(4), attn_bufs(4)
def attn(x):
    q, k, vt, O, B = x
    ops.attn_fwd_log2(q, k, vt, O, None, B, H, S, S, H * 128, S * H * 128)
def attn_one():
    for _ in range(REP): attn(a8)
def attn_two():
    main = torch.cuda.current_stream()
    s1.wait_stream(main); s2.wait_stream(main)
    for _ in range(REP):
        with torch.cuda.stream(s1): attn(a4a)
        with torch.cuda.stream(s2): attn(a4b)
    main.wait_stream(s1); main.wait_stream(s2)
for name, fn in (("attn_B8_one_stream", attn_one), ("attn_2xB4_two_streams", attn_two)):
    t = min(wall(fn, 3) for _ in range(3))
    print(json.dumps({"case": name, "ms": round(t, 3), "tflops": round(4.0 * 8 * H * S * S * 128 * REP / 1e9 / t)}), flush=True)
