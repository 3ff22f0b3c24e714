"""Diagnostic (scratch/libmixgrpo_fstamps.so, built with -DMGX_DIAG_FWD_STAMPS): per-phase cycle shares of a K/V tile of
attn_fwd_kernel.  The stamp sums land in a (deliberately oversized) LSE buffer; behind the LSE values."""
import ctypes as C, math, sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import _lib
h = C.CDLL("scratch/libmixgrpo_fstamps.so")
res, args = _lib.SIGNATURES["mgx_attn_fwd"]
h.mgx_attn_fwd.restype, h.mgx_attn_fwd.argtypes = res, args
torch.manual_seed(0)
B, H, S = 8, 24, 4608
q, k, v = (torch.randn(B, H, S, 128, device="cuda").bfloat16() for _ in range(3))
vt = v.transpose(-1, -2).contiguous()
O = torch.empty(B, S, H * 128, device="cuda", dtype=torch.bfloat16)
nb = (S // 256) * H * B
n_lse = (B * H * S + 1) // 2 * 2
buf = torch.zeros(n_lse + nb * 8 * 8 * 2, dtype=torch.float32, device="cuda")      # [LSE | stamps (int64)]
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    rc = h.mgx_attn_fwd(q.data_ptr(), k.data_ptr(), vt.data_ptr(), O.data_ptr(), buf.data_ptr(), B, H, S, S, H * 128, S * H * 128,
                        1 / math.sqrt(128), st)
    assert rc == 0
torch.cuda.synchronize()
t = buf[n_lse:].view(torch.int64).view(nb, 8, 8)[:, :, :6].double()
names = ["issue staging loads", "S^T (16 MFMA + K reads)", "softmax VALU", "P V (16 MFMA + V reads)", "staging ds_write (+vmcnt)", "barrier"]
tiles = S // 64
tot = t.sum(-1).mean().item()
print(f"cycles per K/V tile (mean over waves): {tot / tiles:.0f}")
for i, n in enumerate(names):
    print(f"  {n:32s} {t[:, :, i].mean().item() / tiles:8.0f}  ({100 * t[:, :, i].mean().item() / tot:5.1f} %)")
for w in range(8):
    print(f"  wave {w}: " + " ".join(f"{t[:, w, i].mean().item() / tiles:7.0f}" for i in range(6)))
