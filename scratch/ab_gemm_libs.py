"""Same-process A/B of the GEMM from two builds of the library (scratch/libmixgrpo_old.so vs the in-tree one): the
handle behind mixgrpo_amd._lib.lib() is swapped between timed loops."""
import ctypes as C, os, sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import _lib, ops
from mixgrpo_amd.ops import Rows
new = _lib.lib()
old = C.CDLL(os.path.join("scratch", "libmixgrpo_old.so"))
for name, (res, args) in _lib.SIGNATURES.items():
    if hasattr(old, name):               # (an older build may lack entry points added since)
        fn = getattr(old, name); fn.restype = res; fn.argtypes = args
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
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = [(36864, 9216, 3072, 0), (36864, 12288, 3072, 1), (36864, 3072, 15360, 2), (32768, 3072, 3072, 2)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for (M, N, K, epi) in shapes:
    fn, C_ = setup(M, N, K, epi)
    _lib._lib = old; fn(); torch.cuda.synchronize(); ref = C_.clone()
    _lib._lib = new; C_.zero_(); fn(); torch.cuda.synchronize()
    same = torch.equal(ref, C_) if epi != 2 else None          # (gate-residual accumulates into C: not comparable this way)
    for _ in range(2): t(fn, 10)
    res = []
    for rep in range(3):
        _lib._lib = old; a = t(fn)
        _lib._lib = new; b = t(fn)
        res.append((a, b))
    fl = 2.0 * M * N * K / 1e9
    print(f"M{M} N{N} K{K} epi{epi} identical={same}: " + "  ".join(f"old {fl/a:.0f} new {fl/b:.0f}" for a, b in res), flush=True)
