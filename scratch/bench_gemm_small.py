"""The txt-stream / small-M GEMM shapes of the training micro-batches (7 and 5 pairs x 512 text tokens, and the B = 1
shared-prefix rollout step): which kernel family serves them better?  MGX_GEMM_BIG_MIN_TILES moves the threshold."""
import os, sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import ops
from mixgrpo_amd.ops import Rows
torch.manual_seed(0)
dev = "cuda"
def bench(M, N, K, epi=0, iters=20):
    A = (torch.randn(M, K, device=dev) * 0.5).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    aux = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi in (1, 4) else None
    gate = torch.ones(1, N, device=dev, dtype=torch.bfloat16) if epi == 2 else None
    kw = dict(aux=aux, gate=gate, gate_ld=N)
    for _ in range(3): ops.gemm(Rows.of(A), W, b, Rows.of(C), N, K, epi, **kw)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gemm(Rows.of(A), W, b, Rows.of(C), N, K, epi, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    print(f"min_tiles={os.environ.get('MGX_GEMM_BIG_MIN_TILES','192'):>4} M{M} N{N} K{K} epi{epi} tiles256={tiles}: {ms:.4f} ms {2*M*N*K/ms/1e9:.0f} TFLOP/s", flush=True)
for M in (3584, 2560, 512, 4608, 4096):
    for (N, K, epi) in ((3072, 3072, 2), (3072, 12288, 2), (9216, 3072, 0), (12288, 3072, 1)):
        bench(M, N, K, epi)
