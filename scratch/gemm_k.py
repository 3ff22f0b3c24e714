import sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import ops
from mixgrpo_amd.ops import Rows
M, N = 36864, 12288
for K in (64, 128, 256, 512, 3072):
    A = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    b = torch.zeros(N, device="cuda", dtype=torch.bfloat16); C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): ops.gemm(Rows.of(A), W, b, Rows.of(C), N, K)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.gemm(Rows.of(A), W, b, Rows.of(C), N, K)
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 10
    print(f"K={K}: {ms:.3f} ms")
