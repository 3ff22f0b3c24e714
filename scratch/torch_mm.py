import torch
M, N, K = 36864, 12288, 3072
A = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
for _ in range(3): torch.matmul(A, W.t())
torch.cuda.synchronize()
