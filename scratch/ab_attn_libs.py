"""Same-process A/B of mgx_attn_fwd from two builds of the library (scratch/libmixgrpo_old.so vs the in-tree one)."""
import ctypes as C, math, os, sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import _lib
new = _lib.lib()
old = C.CDLL(os.path.join("scratch", "libmixgrpo_old.so"))
res, args = _lib.SIGNATURES["mgx_attn_fwd"]
old.mgx_attn_fwd.restype, old.mgx_attn_fwd.argtypes = res, args
torch.manual_seed(0)
B, H, S = 8, 24, 4608
q = torch.randn(B, H, S, 128, device="cuda").bfloat16(); k = torch.randn(B, H, S, 128, device="cuda").bfloat16()
v = torch.randn(B, H, S, 128, device="cuda").bfloat16(); vt = v.transpose(-1, -2).contiguous()
O = torch.empty(B, S, H * 128, device="cuda", dtype=torch.bfloat16); O2 = torch.empty_like(O)
lse = torch.empty(B, H, S, device="cuda"); lse2 = torch.empty_like(lse)
sc = 1 / math.sqrt(128)
st = torch.cuda.current_stream().cuda_stream
def call(lib, O_, l_):
    rc = lib.mgx_attn_fwd(q.data_ptr(), k.data_ptr(), vt.data_ptr(), O_.data_ptr(), l_.data_ptr(), B, H, S, S, H * 128, S * H * 128, sc, st)
    assert rc == 0
def t(lib, n=10):
    call(lib, O, lse); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): call(lib, O, lse)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
call(new, O, lse); call(old, O2, lse2); torch.cuda.synchronize()
print("bit-identical outputs:", torch.equal(O, O2), torch.equal(lse, lse2))
for _ in range(3): t(old, 20)
for rep in range(4):
    print(f"rep {rep}: old {t(old):.3f} ms   new {t(new):.3f} ms", flush=True)
