"""Correctness of mgx_attn_fwd of a (diagnostic but result-preserving) library against an fp32 torch reference."""
import ctypes as C, math, sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import _lib
torch.manual_seed(0)
for path in sys.argv[1:]:
    h = C.CDLL(path)
    res, args = _lib.SIGNATURES["mgx_attn_fwd"]
    h.mgx_attn_fwd.restype, h.mgx_attn_fwd.argtypes = res, args
    for (B, H, S) in [(1, 2, 4608), (2, 3, 333), (1, 1, 64), (1, 2, 200)]:
        Sp = (S + 63) // 64 * 64
        q, k, v = (torch.randn(B, H, S, 128, device="cuda").bfloat16() for _ in range(3))
        vt = torch.zeros(B, H, 128, Sp, device="cuda", dtype=torch.bfloat16); vt[..., :S] = v.transpose(-1, -2)
        O = torch.empty(B, S, H * 128, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B, H, S, device="cuda")
        rc = h.mgx_attn_fwd(q.data_ptr(), k.data_ptr(), vt.data_ptr(), O.data_ptr(), lse.data_ptr(), B, H, S, Sp, H * 128, S * H * 128, 1 / math.sqrt(128), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        sc = (q.float() @ k.float().transpose(-1, -2)) / math.sqrt(128)
        ref = (torch.softmax(sc, -1) @ v.float()).transpose(1, 2).reshape(B, S, H * 128)
        print(path, (B, H, S), "rc", rc, "rel err", ((O.float() - ref).norm() / ref.norm()).item(), "lse err", (lse - torch.logsumexp(sc, -1)).abs().max().item(), flush=True)
