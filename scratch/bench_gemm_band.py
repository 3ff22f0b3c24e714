"""Tile-order sweep (MGX_GEMM_BAND, one process per value) of the ping-pong GEMM on the dominant shapes."""
import os, statistics, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, ".")
    from mixgrpo_amd import ops
    from mixgrpo_amd.ops import Rows
    torch.manual_seed(0)
    def t(fn, n=10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    out = []
    for (M, N, K, epi) in [(36864, 9216, 3072, 0), (36864, 12288, 3072, 1), (36864, 3072, 15360, 2), (36864, 3072, 3072, 2), (21504, 3072, 32256, 3)]:
        A = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        b = (torch.randn(N, device="cuda") * 0.1).bfloat16()
        C_ = torch.zeros(M, N, device="cuda", dtype=torch.float32 if epi == 3 else torch.bfloat16)
        aux = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if epi in (1, 4) else None
        gate = torch.ones(1, N, device="cuda", dtype=torch.bfloat16) if epi == 2 else None
        fn = lambda: ops.gemm(Rows.of(A), W, None if epi == 3 else b, Rows.of(C_), N, K, epi, aux=aux, gate=gate, gate_ld=N, beta=1.0 if epi == 3 else 0.0)
        for _ in range(3): t(fn, 5)
        out.append(f"{2.0 * M * N * K / 1e9 / statistics.median([t(fn) for _ in range(5)]):.0f}")
    print("band", os.environ.get("MGX_GEMM_BAND", "rule"), " ".join(out), flush=True)
else:
    print("shapes: N9216/K3072 epi0 | N12288/K3072 epi1 | N3072/K15360 epi2 | N3072/K3072 epi2 | M21504 N3072 K32256 epi3   [TFLOP/s]", flush=True)
    for band in ["", "1", "2", "3", "4", "6", "8", "12", ""]:
        env = dict(os.environ, MGX_GEMM_MODE="9")
        if band: env["MGX_GEMM_BAND"] = band
        subprocess.run([sys.executable, __file__, "child"], env=env, check=True)
