"""Soak screen for the ping-pong GEMM's LDS hazards: FLUX-size GEMMs on exactly representable operands, every output element
compared bit for bit with the fp32 reference on every launch (tests/test_hip_gemm.py runs 4 launches; this runs hundreds,
with other kernels interleaved so that the memory system's timing varies)."""
import sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import ops
from mixgrpo_amd.ops import Rows
torch.manual_seed(0)
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 200
junk = torch.randn(64 * 1024 * 1024, device="cuda")
for (M, N, K) in [(36864, 3072, 3072), (36864, 9216, 3072), (36864, 3072, 15360), (32256, 12288, 3072), (4096, 3072, 12288)]:
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = (torch.randint(-4, 5, (M, K), generator=g, device="cuda").float() / 8).bfloat16()
    W = (torch.randint(-2, 3, (N, K), generator=g, device="cuda").float() / 4).bfloat16()
    ref = torch.empty(M, N, dtype=torch.float32, device="cuda")
    for r0 in range(0, M, 8192):
        ref[r0:r0 + 8192] = A[r0:r0 + 8192].float() @ W.float().t()
    C = torch.empty(M, N, dtype=torch.float32, device="cuda")
    bad = 0
    for it in range(n_iter):
        if it % 3 == 1:
            junk.mul_(1.0001)                       # a bandwidth-bound neighbour right before the launch
        ops.gemm(Rows.of(A), W, None, Rows.of(C), N, K, 3, beta=0.0)
        if not torch.equal(C, ref):
            bad += 1
    print(f"M{M} N{N} K{K}: {n_iter} launches, {bad} with a wrong element", flush=True)
