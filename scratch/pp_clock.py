"""Diagnostic (scratch/libmixgrpo_ppclock.so, -DMGX_DIAG_PP_CLOCK): in-kernel clock of the ping-pong GEMM =
d(s_memtime) / d(s_memrealtime) x 100 MHz per workgroup (MI355X_MICROARCH.md 'DVFS give-back' item 6), after >= 2 s of
back-to-back launches on random data, and shader cycles per K-tile (matrix-pipe floor: 2048 = 128 MFMA x 16 cycles per SIMD)."""
import ctypes as C, os, sys, time, torch
sys.path.insert(0, ".")
from mixgrpo_amd import _lib
h = C.CDLL(os.environ.get("PP_LIB", "scratch/libmixgrpo_ppclock.so"))
res, args = _lib.SIGNATURES["mgx_gemm_bf16"]
h.mgx_gemm_bf16.restype, h.mgx_gemm_bf16.argtypes = res, args
torch.manual_seed(0)
for (M, N, K) in [(36864, 9216, 3072), (36864, 3072, 15360)]:
    A = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    b = torch.zeros(N, device="cuda", dtype=torch.bfloat16); Cm = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    dbg = torch.zeros(256 * 2, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    def go():
        rc = h.mgx_gemm_bf16(A.data_ptr(), W.data_ptr(), b.data_ptr(), Cm.data_ptr(), None, dbg.data_ptr(), N, M, N, K, K, 1 << 40, 0, K, N, 1 << 40, 0, 0, 0, 0.0, st)
        assert rc == 0
    t0 = time.time()
    while time.time() - t0 < 2.5:
        for _ in range(50): go()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): go()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    d = dbg.view(256, 2).double()
    clk = (d[:, 0] / d[:, 1] * 0.1).median().item()
    tiles_per_wg = (M // 256) * (N // 256) / 256
    cyc = (d[:, 0] / (tiles_per_wg * (K // 64))).median().item()
    tf = 2.0 * M * N * K / ms / 1e9
    print(f"M{M} N{N} K{K}: {tf:.0f} TFLOP/s, in-kernel clock {clk:.3f} GHz, {cyc:.0f} cycles per K-tile (incl. epilogue share), "
          f"matrix pipe busy {2048 / cyc:.3f}, peak at this clock {2500 * clk / 2.4:.0f} TFLOP/s", flush=True)
