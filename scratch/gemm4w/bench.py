"""Scratch: the 4-wave prototype K-loop (scratch/gemm4w) against the shipped ping-pong GEMM on FLUX shapes, same process,
alternating arms; numerics of the prototype checked against torch on the first shape."""
import ctypes, os, sys, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from mixgrpo_amd import ops
from mixgrpo_amd.ops import Rows
lib = ctypes.CDLL(os.path.join(HERE, "gemm4w.so"))
lib.gemm4w.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
torch.manual_seed(0)
st = torch.cuda.current_stream().cuda_stream

def check():
    M, N, K = 512, 768, 640
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    assert lib.gemm4w(A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, st) == 0
    torch.cuda.synchronize()
    ref = A.float() @ W.float().t()
    rel = ((C.float() - ref).norm() / ref.norm()).item()
    print(f"prototype vs torch at {M}x{N}x{K}: rel {rel:.2e}", flush=True)
    assert rel < 4e-3

check()
shapes = [(36864, 3072, 15360), (36864, 3072, 12288), (36864, 9216, 3072), (36864, 12288, 3072), (36864, 3072, 3072), (32256, 3072, 15360)]
for (M, N, K) in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    b = torch.zeros(N, device="cuda", dtype=torch.bfloat16)
    C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    def arm_new(): lib.gemm4w(A.data_ptr(), W.data_ptr(), C.data_ptr(), M, N, K, st)
    def arm_old(): ops.gemm(Rows.of(A), W, b, Rows.of(C), N, K, 0)
    res = {"old": [], "new": []}
    for rnd in range(3):
        for name, fn in (("old", arm_old), ("new", arm_new)):
            for _ in range(5): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / 20)
    fl = 2.0 * M * N * K
    print(f"M {M} N {N} K {K}: shipped {min(res['old']):.4f} ms {fl / min(res['old']) / 1e9:.0f} TF | 4-wave prototype {min(res['new']):.4f} ms {fl / min(res['new']) / 1e9:.0f} TF", flush=True)
