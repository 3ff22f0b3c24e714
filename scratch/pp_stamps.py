"""Diagnostic (scratch/libmixgrpo_ppstamps.so, -DMGX_DIAG_PP_STAMPS): cycles per [section + barrier wait] of the ping-pong
GEMM's K-loop, per wave half.  Section k of a wave = what follows its k-th barrier: early half MA, LB, MB, LA(next); the late
half runs the same text one barrier later.  NOTE the stamps are intrusive: every barrier gains an s_memtime + lgkmcnt(0)
(~100 cycles), so only differences between sections / builds mean anything; scratch/pp_clock.py gives the undisturbed
cycles per K-tile.  (profiles/r02_pp_stamps.log is the same script on the earlier four-phase loop: sections M0 L1 M1 L2 M2 L3 M3 L0.)"""
import ctypes as C, os, sys, torch
os.environ.pop("MGX_GEMM_MODE", None)          # the ping-pong kernel is the default
sys.path.insert(0, ".")
from mixgrpo_amd import _lib
LIB = os.environ.get("PP_LIB", "scratch/libmixgrpo_ppstamps.so")
h = C.CDLL(LIB)
res, args = _lib.SIGNATURES["mgx_gemm_bf16"]
h.mgx_gemm_bf16.restype, h.mgx_gemm_bf16.argtypes = res, args
torch.manual_seed(0)
M, N, K = 36864, 9216, 3072
if len(sys.argv) > 3: M, N, K = (int(v) for v in sys.argv[1:4])
A = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
b = torch.zeros(N, device="cuda", dtype=torch.bfloat16); Cm = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
dbg = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    rc = h.mgx_gemm_bf16(A.data_ptr(), W.data_ptr(), b.data_ptr(), Cm.data_ptr(), None, dbg.data_ptr(), N, M, N, K, K, 1 << 40, 0, K, N, 1 << 40, 0, 0, 0, 0.0, st)
    assert rc == 0, _lib.lib().mgx_last_error()
torch.cuda.synchronize()
t = dbg.view(256, 8, 8).double()
tiles_per_wg = (M // 256) * (N // 256) / 256
kt = K // 64
per = t / (tiles_per_wg * kt) * 2          # the 8 counters cover two K-tiles
names_e = ["MA", "LB", "MB", "LA'"] * 2
div = 2
print(f"{LIB} M{M} N{N} K{K}: cycles per K-tile: early half {per[:, :4].sum(-1).mean() / div:.0f}, late half {per[:, 4:].sum(-1).mean() / div:.0f}")
print("section (early half naming) : early half | late half   [cycles incl. the barrier wait that ends it]")
for k in range(8):
    print(f"  after barrier {k}: {names_e[k]:4s} {per[:, :4, k].mean():7.0f} | {per[:, 4:, k].mean():7.0f}")
