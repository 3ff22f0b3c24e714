"""Stream-K tail A/B: the same shapes with and without the workspace, arms alternating in one process."""
import json, os, sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import ops
from mixgrpo_amd.ops import Rows
torch.manual_seed(0)
dev = "cuda"
SHAPES = [  # (M, N, K, epi)
    (36864, 3072, 15360, 2), (36864, 9216, 3072, 0), (36864, 12288, 3072, 1), (32768, 3072, 12288, 2), (32768, 3072, 3072, 2),
    (4096, 3072, 12288, 2), (4096, 9216, 3072, 0), (4096, 3072, 3072, 2), (4608, 12288, 3072, 1), (4608, 3072, 15360, 2),
    (12288, 3072, 28672, 3), (3072, 12288, 28672, 3), (9216, 3072, 28672, 3), (3072, 3072, 28672, 3), (3072, 15360, 32256, 3),
    (21504, 3072, 32256, 3), (28672, 3072, 12288, 2), (28672, 3072, 3072, 2), (23040, 3072, 15360, 2), (23040, 3072, 21504, 0),
    # micro-batch 4
    (18432, 3072, 15360, 2), (18432, 9216, 3072, 0), (18432, 12288, 3072, 1), (18432, 3072, 21504, 0), (16384, 3072, 12288, 2),
    (16384, 3072, 3072, 2), (16384, 9216, 3072, 0), (2048, 3072, 12288, 2), (2048, 12288, 3072, 1), (2048, 9216, 3072, 0),
    (12288, 3072, 16384, 3), (3072, 12288, 16384, 3), (9216, 3072, 16384, 3), (3072, 3072, 16384, 3), (3072, 15360, 18432, 3),
    (21504, 3072, 18432, 3),
]
def setup(M, N, K, epi):
    A = (torch.randn(M, K, device=dev) * 0.5).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    b = None if epi == 3 else torch.zeros(N, device=dev, dtype=torch.bfloat16)
    C = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi == 3 else torch.bfloat16)
    aux = torch.randn(M, N, device=dev).bfloat16() if epi in (1, 4) else None
    gate = torch.ones(1, N, device=dev, dtype=torch.bfloat16) if epi == 2 else None
    return A, W, b, C, dict(aux=aux, gate=gate, gate_ld=N, beta=1.0 if epi == 3 else 0.0)
def run(A, W, b, C, kw, N, K, epi, iters):
    for _ in range(2): ops.gemm(Rows.of(A), W, b, Rows.of(C), N, K, epi, **kw)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gemm(Rows.of(A), W, b, Rows.of(C), N, K, epi, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
only = os.environ.get("SK_ONLY")
for (M, N, K, epi) in SHAPES:
    t = setup(M, N, K, epi)
    res = {}
    for rep in range(2):
        for sk in (False, True):
            ops.GEMM_STREAM_K = sk
            ms = run(*t, N, K, epi, 8)
            res[sk] = min(res.get(sk, 1e9), ms)
    tiles = -(-M // 256) * -(-N // 256)
    fl = 2.0 * M * N * K
    print(json.dumps(dict(M=M, N=N, K=K, epi=epi, rounds=round(tiles / 256, 3), ms_off=round(res[False], 4), ms_sk=round(res[True], 4),
                          tf_off=round(fl / res[False] / 1e9), tf_sk=round(fl / res[True] / 1e9), gain=round(res[False] / res[True], 4))), flush=True)
    del t
