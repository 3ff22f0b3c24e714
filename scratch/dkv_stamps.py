"""Diagnostic (scratch/libmixgrpo_stamps.so, built with -DMGX_DIAG_DKV_STAMPS): per-phase cycle shares of a q-tile of
attn_bwd_dkv_kernel.  The stamp sums land in the dQ buffer (the dq kernel is not launched in that build)."""
import ctypes as C, math, sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import _lib
h = C.CDLL("scratch/libmixgrpo_stamps.so")
res, args = _lib.SIGNATURES["mgx_attn_bwd"]
h.mgx_attn_bwd.restype, h.mgx_attn_bwd.argtypes = res, args
torch.manual_seed(0)
B, H, S = 8, 24, 4608
q, k, v = (torch.randn(B, H, S, 128, device="cuda").bfloat16() for _ in range(3))
qt, kt = q.transpose(-1, -2).contiguous(), k.transpose(-1, -2).contiguous()
O = torch.randn(B, S, H * 128, device="cuda").bfloat16(); do = torch.randn_like(O)
lse = torch.randn(B, H, S, device="cuda") + 8
delta = torch.empty(B, H, S, device="cuda"); dOt = torch.zeros(B, H, 128, S, device="cuda", dtype=torch.bfloat16)
dQ, dK, dV = (torch.zeros_like(q) for _ in range(3))
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    rc = h.mgx_attn_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), qt.data_ptr(), kt.data_ptr(), O.data_ptr(), do.data_ptr(), lse.data_ptr(),
                        delta.data_ptr(), dOt.data_ptr(), dQ.data_ptr(), dK.data_ptr(), dV.data_ptr(), B, H, S, S, H * 128, S * H * 128,
                        1 / math.sqrt(128), st)
    assert rc == 0
torch.cuda.synchronize()
nb = (S // 256) * H * B
t = dQ.view(torch.int64).flatten()[:nb * 64].view(nb, 8, 8)[:, :, :6].double()
names = ["issue staging loads", "S / dP (16 MFMA + reads)", "exp / dS VALU", "dV / dK (16 MFMA + reads)", "staging ds_write (+vmcnt)", "barrier"]
tiles = S // 32
tot = t.sum(-1).mean().item()
print(f"cycles per q-tile (mean over waves): {tot / tiles:.0f}")
for i, n in enumerate(names):
    print(f"  {n:32s} {t[:, :, i].mean().item() / tiles:8.0f}  ({100 * t[:, :, i].mean().item() / tot:5.1f} %)")
for w in range(8):
    print(f"  wave {w}: " + " ".join(f"{t[:, w, i].mean().item() / tiles:7.0f}" for i in range(6)))
