import sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import ops
from mixgrpo_amd.ops import Rows
M, N, K = 36864, 12288, 3072
A = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
b = torch.zeros(N, device="cuda", dtype=torch.bfloat16); C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(5): ops.gemm(Rows.of(A), W, b, Rows.of(C), N, K)
torch.cuda.synchronize()
