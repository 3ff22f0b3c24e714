"""Scratch A/B: gemm_pp_kernel's ping-pong K-loop on v_mfma_f32_16x16x32_bf16 (shipped) against the same
loop on v_mfma_f32_32x32x16_bf16 -- same tiles, LDS images, DMA pieces, barriers, tile walk; plain bf16 epilogue in both arms
(scratch/gemm_mfma32/kernel.hip).  Real operands, the six dominant shapes of the step, arms alternating in one process;
both arms are checked against torch first.  The product kernel (same shapes, bias epilogue) runs as a third arm for reference."""
import ctypes, os, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from mixgrpo_amd import ops
from mixgrpo_amd.ops import Rows
lib = ctypes.CDLL(os.path.join(HERE, "gemm_mfma32.so"))
lib.pp_gemm.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
torch.manual_seed(0)
st = torch.cuda.current_stream().cuda_stream


def check():
    for (M, N, K) in ((1024, 1280, 640), (2048, 4096, 320), (8192, 9216, 3072)):
        A = torch.randn(M, K, device="cuda").bfloat16()
        W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        ref = A.float() @ W.float().t()
        for arm in (16, 32):
            C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
            rc = lib.pp_gemm(arm, A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, st)
            assert rc == 0, rc
            torch.cuda.synchronize()
            rel = ((C.float() - ref).norm() / ref.norm()).item()
            print(f"arm {arm} vs torch at {M}x{N}x{K}: rel {rel:.2e}", flush=True)
            assert rel < 4e-3, (arm, rel)


check()
shapes = [(36864, 3072, 15360), (36864, 3072, 12288), (36864, 9216, 3072), (36864, 12288, 3072), (32768, 3072, 3072),
          (32256, 3072, 15360), (16384, 3072, 12288), (3072, 12288, 32768)]
tot = {16: 0.0, 32: 0.0}
for (M, N, K) in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    b = torch.zeros(N, device="cuda", dtype=torch.bfloat16)
    C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    arms = {16: lambda: lib.pp_gemm(16, A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, st),
            32: lambda: lib.pp_gemm(32, A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, st),
            "product": lambda: ops.gemm(Rows.of(A), W, b, Rows.of(C), N, K, 0)}
    res = {k: [] for k in arms}
    for rnd in range(3):
        for name, fn in arms.items():
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / 20)
    fl = 2.0 * M * N * K
    t16, t32, tp = min(res[16]), min(res[32]), min(res["product"])
    tot[16] += t16
    tot[32] += t32
    print(f"M {M} N {N} K {K}: 16x16x32 {t16:.4f} ms {fl / t16 / 1e9:.0f} TF | 32x32x16 {t32:.4f} ms {fl / t32 / 1e9:.0f} TF "
          f"({100 * (t16 / t32 - 1):+.2f} %) | product {tp:.4f} ms {fl / tp / 1e9:.0f} TF", flush=True)
print(f"sum over the shapes: 16x16x32 {tot[16]:.3f} ms, 32x32x16 {tot[32]:.3f} ms ({100 * (tot[16] / tot[32] - 1):+.2f} %)", flush=True)
