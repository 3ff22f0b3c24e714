"""Times mgx_attn_fwd of a (diagnostic) library at B 8, H 24, S 4608: python scratch/attn_fwd_time.py scratch/libX.so ..."""
import ctypes as C, math, sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import _lib
torch.manual_seed(0)
B, H, S = 8, 24, 4608
q, k, v = (torch.randn(B, H, S, 128, device="cuda").bfloat16() for _ in range(3))
vt = v.transpose(-1, -2).contiguous()
O = torch.empty(B, S, H * 128, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B, H, S, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for path in sys.argv[1:] * 2:
    h = C.CDLL(path)
    res, args = _lib.SIGNATURES["mgx_attn_fwd"]
    h.mgx_attn_fwd.restype, h.mgx_attn_fwd.argtypes = res, args
    fn = lambda: h.mgx_attn_fwd(q.data_ptr(), k.data_ptr(), vt.data_ptr(), O.data_ptr(), lse.data_ptr(), B, H, S, S, H * 128, S * H * 128, 1 / math.sqrt(128), st)
    for _ in range(10): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ms = sorted(ts)[2]
    print(f"{path}: {ms:.3f} ms {4.0 * B * H * S * S * 128 / ms / 1e9:.0f} TFLOP/s", flush=True)
