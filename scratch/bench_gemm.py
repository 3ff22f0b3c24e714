import sys, time, torch
sys.path.insert(0, ".")
from mixgrpo_amd import ops
from mixgrpo_amd.ops import Rows
torch.manual_seed(0)
dev = "cuda"
def check(M, N, K, epi=0):
    A = (torch.randn(M, K, device=dev) * 0.5).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    b = (torch.randn(N, device=dev) * 0.1).bfloat16()
    C = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    ops.gemm(Rows.of(A), W, b, Rows.of(C), N, K, epi)
    ref = (A.float() @ W.float().t() + b.float()).bfloat16()
    if epi == 1: ref = torch.nn.functional.gelu(ref.float(), approximate="tanh").bfloat16()
    err = (C.float() - ref.float()).abs().max().item()
    nbad = (C != ref).sum().item()
    print(f"check M{M} N{N} K{K} epi{epi}: maxerr {err:.4g} mismatching {nbad}/{C.numel()}")
for shp in [(128,128,64),(256,256,128),(200,132,192),(1000,64,3072),(4608,3072,3072)]:
    check(*shp)
check(512, 1024, 256, 1)
def bench(M, N, K, iters=10):
    A = (torch.randn(M, K, device=dev) * 0.5).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    b = torch.zeros(N, device=dev, dtype=torch.bfloat16); C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(2): ops.gemm(Rows.of(A), W, b, Rows.of(C), N, K)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gemm(Rows.of(A), W, b, Rows.of(C), N, K)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    t0=time.time()
    for _ in range(3): torch.matmul(A, W.t())
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): torch.matmul(A, W.t())
    e1.record(); torch.cuda.synchronize(); ms2 = e0.elapsed_time(e1) / iters
    print(f"bench M{M} N{N} K{K}: {ms:.3f} ms {2*M*N*K/ms/1e9:.0f} TFLOP/s | torch(hipblaslt) {ms2:.3f} ms {2*M*N*K/ms2/1e9:.0f} TFLOP/s")
for shp in [(4608,9216,3072),(36864,9216,3072),(36864,12288,3072),(36864,3072,12288),(36864,3072,15360),(36864,3072,3072)]:
    bench(*shp)
