"""transpose8_kernel A/B: in-tree library vs scratch/libmixgrpo_tr16x4.so (the old 16 x 4 thread map); checks the result."""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import _lib
torch.manual_seed(0)
libs = {"new (8x8 per wave)": _lib.lib(), "old (16x4 per wave)": C.CDLL("scratch/libmixgrpo_tr16x4.so")}
res, args = _lib.SIGNATURES["mgx_transpose_bf16"]
for h in libs.values():
    h.mgx_transpose_bf16.restype, h.mgx_transpose_bf16.argtypes = res, args
st = torch.cuda.current_stream().cuda_stream
for (M, N) in [(32256, 12288), (32256, 3072), (32256, 9216), (21504, 3072), (4096, 3072)]:
    x = torch.randn(M, N, device="cuda").bfloat16()
    out = torch.empty(N, M, device="cuda", dtype=torch.bfloat16)
    part = torch.empty((M + 63) // 64 * N, device="cuda"); cs = torch.zeros(N, device="cuda")
    line = f"M{M} N{N}:"
    for name, h in list(libs.items()) * 2:
        for with_cs in (False, True):
            fn = lambda: h.mgx_transpose_bf16(x.data_ptr(), out.data_ptr(), part.data_ptr() if with_cs else None, cs.data_ptr() if with_cs else None, 0.0, M, N, N, 1 << 40, 0, M, st)
            out.zero_(); assert fn() == 0; torch.cuda.synchronize()
            assert torch.equal(out, x.t())
            if with_cs: assert torch.allclose(cs, x.float().sum(0), rtol=1e-4, atol=1e-2)
            for _ in range(5): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            line += f"  {name.split()[0]}{'+colsum' if with_cs else ''} {ms * 1e3:.0f} us {4.0 * M * N / ms / 1e9:.2f} TB/s"
    print(line, flush=True)
