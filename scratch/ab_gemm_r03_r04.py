"""Same-process A/B of `mgx_gemm_bf16` (no stream-K workspace: every tile whole) from the round-3 library
(scratch/libmixgrpo_old.so, built from git 0ebf66d) and the in-tree one: did the unit walk that the stream-K tail added to
gemm_pp_kernel cost the plain path anything?  Arms alternate; the last column is the in-tree library WITH the workspace."""
import ctypes as C, os, sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import _lib
new = _lib.lib()
old = C.CDLL(os.path.join("scratch", "libmixgrpo_old.so"))
res, args = _lib.SIGNATURES["mgx_gemm_bf16"]
old.mgx_gemm_bf16.restype = res
old.mgx_gemm_bf16.argtypes = args
torch.manual_seed(0)
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
ws = torch.empty(new.mgx_gemm_sk_workspace_elems(), dtype=torch.float32, device=dev)


def setup(M, N, K, epi):
    A = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    b = None if epi == 3 else torch.zeros(N, device=dev, dtype=torch.bfloat16)
    Cm = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi == 3 else torch.bfloat16)
    aux = torch.randn(M, N, device=dev).bfloat16() if epi in (1, 4) else None
    gate = torch.ones(1, N, device=dev, dtype=torch.bfloat16) if epi == 2 else None
    p = lambda t: None if t is None else t.data_ptr()
    base = (p(A), p(W), p(b), p(Cm), p(gate), p(aux), N, M, N, K, K, 1 << 40, 0, K, N, 1 << 40, 0, N, epi, 1.0 if epi == 3 else 0.0)
    return (lambda h: h.mgx_gemm_bf16(*base, st)), (lambda: new.mgx_gemm_bf16_sk(*base, ws.data_ptr(), ws.numel(), st)), (A, W, b, Cm, aux, gate)


def t(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


shapes = [(36864, 3072, 15360, 2), (36864, 9216, 3072, 0), (36864, 12288, 3072, 1), (32768, 3072, 12288, 2), (32768, 3072, 3072, 2),
          (18432, 3072, 21504, 0), (21504, 3072, 18432, 3), (18432, 12288, 3072, 4), (3072, 12288, 16384, 3), (4096, 3072, 12288, 2)]
tot = [0.0, 0.0, 0.0]
for (M, N, K, epi) in shapes:
    plain, sk, keep = setup(M, N, K, epi)
    for _ in range(2):
        t(lambda: plain(old), 5), t(lambda: plain(new), 5)
    r = [[], [], []]
    for rep in range(3):
        r[0].append(t(lambda: plain(old)))
        r[1].append(t(lambda: plain(new)))
        r[2].append(t(sk))
    a, b_, c = (min(x) for x in r)
    fl = 2.0 * M * N * K / 1e9
    tot[0] += a; tot[1] += b_; tot[2] += c
    print(f"M{M} N{N} K{K} epi{epi}: r03 {a:.4f} ms {fl / a:.0f} TF | r04 plain {b_:.4f} ms {fl / b_:.0f} TF ({100 * (a / b_ - 1):+.2f} %) | "
          f"r04 stream-K {c:.4f} ms {fl / c:.0f} TF ({100 * (a / c - 1):+.2f} %)", flush=True)
    del keep
print(f"sum: r03 {tot[0]:.3f} ms | r04 plain {tot[1]:.3f} ms ({100 * (tot[0] / tot[1] - 1):+.2f} %) | r04 stream-K {tot[2]:.3f} ms "
      f"({100 * (tot[0] / tot[2] - 1):+.2f} %)", flush=True)
