"""One GEMM shape, a few launches: the target of the rocprofv3 PMC passes (profiles/README.md).
usage: python scratch/gemm_one.py M N K epi [iters]"""
import sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import ops
from mixgrpo_amd.ops import Rows
M, N, K, epi = (int(v) for v in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
A = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
b = None if epi == 3 else torch.zeros(N, device="cuda", dtype=torch.bfloat16)
C = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi == 3 else torch.bfloat16)
gate = torch.ones(1, N, device="cuda", dtype=torch.bfloat16) if epi == 2 else None
for _ in range(iters): ops.gemm(Rows.of(A), W, b, Rows.of(C), N, K, epi, gate=gate, gate_ld=N, beta=1.0 if epi == 3 else 0.0)
torch.cuda.synchronize()
