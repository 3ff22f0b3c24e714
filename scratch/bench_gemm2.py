import os, sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import ops
from mixgrpo_amd.ops import Rows
torch.manual_seed(0)
dev = "cuda"
def bench(M, N, K, epi=0, iters=8):
    A = (torch.randn(M, K, device=dev) * 0.5).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    C = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi == 3 else torch.bfloat16)
    aux = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi in (1, 4) else None
    gate = torch.ones(1, N, device=dev, dtype=torch.bfloat16) if epi == 2 else None
    kw = dict(aux=aux, gate=gate, gate_ld=N, beta=1.0 if epi == 3 else 0.0)
    for _ in range(2): ops.gemm(Rows.of(A), W, None if epi == 3 else b, Rows.of(C), N, K, epi, **kw)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gemm(Rows.of(A), W, None if epi == 3 else b, Rows.of(C), N, K, epi, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"tag={os.environ.get('MGX_BENCH_TAG','0'):>3} M{M} N{N} K{K} epi{epi}: {ms:.3f} ms {2*M*N*K/ms/1e9:.0f} TFLOP/s", flush=True)
shapes = [(36864,9216,3072,0),(36864,12288,3072,1),(36864,3072,15360,2),(32768,3072,3072,2),(32768,3072,12288,2),
          (27648,9216,3072,0),(27648,12288,3072,1),(27648,3072,15360,2),(27648,3072,21504,0),(27648,12288,3072,4),
          (21504,3072,27648,3),(4096,9216,3072,0),(4096,12288,3072,1)]
for s in shapes: bench(*s)
