"""Times the dominant GEMM shapes with the kernel family selected by MGX_GEMM_MODE (unset = the ping-pong kernel, 6 = the
persistent kernel of round 1; in the logs of 2026-10: 8 = four-phase ping-pong, 9 = two-phase ping-pong = today's default);
checks the result against torch.matmul on one shape."""
import os, statistics, sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import ops
from mixgrpo_amd.ops import Rows
torch.manual_seed(0)
dev = "cuda"
def setup(M, N, K, epi):
    A = (torch.randn(M, K, device=dev) * 0.5).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    b = (torch.randn(N, device=dev) * 0.1).bfloat16()
    C_ = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi == 3 else torch.bfloat16)
    aux = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi in (1, 4) else None
    gate = torch.ones(1, N, device=dev, dtype=torch.bfloat16) if epi == 2 else None
    kw = dict(aux=aux, gate=gate, gate_ld=N, beta=1.0 if epi == 3 else 0.0)
    return (lambda: ops.gemm(Rows.of(A), W, None if epi == 3 else b, Rows.of(C_), N, K, epi, **kw)), A, W, b, C_
def t(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
fn, A, W, b, C_ = setup(4096 + 256, 3072, 3072, 0)
fn(); torch.cuda.synchronize()
ref = (A.float() @ W.float().t() + b.float()).bfloat16()
print("mode", os.environ.get("MGX_GEMM_MODE", "default"), "max |err| vs fp32 matmul:", (C_.float() - ref.float()).abs().max().item())
for (M, N, K, epi) in [(36864, 9216, 3072, 0), (36864, 12288, 3072, 1), (36864, 3072, 15360, 2), (32768, 3072, 3072, 2), (21504, 3072, 32256, 3), (32256, 12288, 3072, 1)]:
    fn, *_ = setup(M, N, K, epi)
    for _ in range(3): t(fn, 5)
    v = [t(fn) for _ in range(5)]
    print(f"M{M} N{N} K{K} epi{epi}: {2.0 * M * N * K / 1e9 / statistics.median(v):.0f} TFLOP/s", flush=True)
