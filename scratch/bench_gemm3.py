"""Three K = 3072 shapes + one long-K shape, repeated: for timing-only A/B builds of the epilogue."""
import os, sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import ops
from mixgrpo_amd.ops import Rows
torch.manual_seed(0)
dev = "cuda"
def bench(M, N, K, epi=0, iters=10):
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
    print(f"tag={os.environ.get('MGX_BENCH_TAG','0'):>8} M{M} N{N} K{K} epi{epi}: {ms:.3f} ms {2*M*N*K/ms/1e9:.0f} TFLOP/s", flush=True)
for rep in range(2):
    for s in [(36864,9216,3072,0),(36864,12288,3072,1),(32768,3072,3072,2),(36864,3072,15360,2)]: bench(*s)
