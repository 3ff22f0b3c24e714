"""Same-process A/B of the GEMM from SEVERAL builds of the library: `python scratch/ab_gemm_multi.py name=path ...`
(the in-tree library is always included as `tree`).  Interleaved rounds, median TFLOP/s per (shape, build)."""
import ctypes as C, os, statistics, sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import _lib, ops
from mixgrpo_amd.ops import Rows
libs = {"tree": _lib.lib()}
for a in sys.argv[1:]:
    n, pth = a.split("=")
    h = C.CDLL(pth)
    for name, (res, args) in _lib.SIGNATURES.items():
        if hasattr(h, name):
            fn = getattr(h, name); fn.restype = res; fn.argtypes = args
    libs[n] = h
torch.manual_seed(0)
dev = "cuda"
def setup(M, N, K, epi):
    A = (torch.randn(M, K, device=dev) * 0.5).bfloat16(); W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    C_ = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi == 3 else torch.bfloat16)
    aux = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if epi in (1, 4) else None
    gate = torch.ones(1, N, device=dev, dtype=torch.bfloat16) if epi == 2 else None
    kw = dict(aux=aux, gate=gate, gate_ld=N, beta=1.0 if epi == 3 else 0.0)
    return lambda: ops.gemm(Rows.of(A), W, None if epi == 3 else b, Rows.of(C_), N, K, epi, **kw), C_
def t(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = [(36864, 9216, 3072, 0), (36864, 12288, 3072, 1), (36864, 3072, 15360, 2), (32768, 3072, 3072, 2), (21504, 3072, 32256, 3)]
for (M, N, K, epi) in shapes:
    fn, C_ = setup(M, N, K, epi)
    ref = None
    same = {}
    for n, h in libs.items():
        _lib._lib = h; C_.zero_(); fn(); torch.cuda.synchronize()
        if ref is None: ref = C_.clone()
        same[n] = bool(torch.equal(ref, C_))
    for n, h in libs.items():
        _lib._lib = h
        for _ in range(2): t(fn, 5)
    res = {n: [] for n in libs}
    for rep in range(5):
        for n, h in libs.items():
            _lib._lib = h
            res[n].append(t(fn))
    fl = 2.0 * M * N * K / 1e9
    print(f"M{M} N{N} K{K} epi{epi}: " + "  ".join(f"{n} {fl / statistics.median(v):.0f} (min {fl / max(v):.0f}, same={same[n]})" for n, v in res.items()), flush=True)
_lib._lib = libs["tree"]
