"""Scratch A/B: start stagger of the persistent GEMM's XCDs (MGX_GEMM_STAGGER, read per launch in the experimental build):
same process, alternating arms, several FLUX shapes."""
import os, sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import ops
from mixgrpo_amd.ops import Rows
shapes = [(36864, 9216, 3072, 0), (36864, 12288, 3072, 1), (36864, 3072, 3072, 2), (36864, 3072, 15360, 2), (36864, 3072, 12288, 2),
          (32256, 12288, 3072, 1), (21504, 3072, 32256, 3)]
arms = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,4,8")]
torch.manual_seed(0)
for (M, N, K, epi) in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    C = (torch.randn(M, N, device="cuda").bfloat16() if epi != 3 else torch.zeros(M, N, device="cuda"))
    gate = torch.randn(8, N, device="cuda").bfloat16() if epi == 2 else None
    aux = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if epi == 1 else None
    def run():
        if epi == 2:
            ops.gemm(Rows(A, M, K, M // 8, (M // 8) * K), W, b, Rows(C, M, N, M // 8, (M // 8) * N), N, K, 2, gate=gate, gate_ld=N)
        elif epi == 3:
            ops.gemm(Rows.of(A), W, None, Rows.of(C), N, K, 3, beta=1.0)
        else:
            ops.gemm(Rows.of(A), W, b, Rows.of(C), N, K, epi, aux=aux)
    res = {a: [] for a in arms}
    for rnd in range(3):
        for a in arms:
            os.environ["MGX_GEMM_STAGGER"] = str(a)
            for _ in range(5): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): run()
            e1.record(); torch.cuda.synchronize()
            res[a].append(e0.elapsed_time(e1) / 30)
    line = f"M {M} N {N} K {K} epi {epi}: " + "  ".join(f"s{a}: {min(v):.4f} ms {2.0 * M * N * K / min(v) / 1e9:.0f} TF" for a, v in res.items())
    print(line, flush=True)
