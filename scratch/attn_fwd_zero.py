"""Is the forward attention clock-limited?  Same kernel on random and on all-zero operands (MI355X_MICROARCH.md, DVFS give-back 1)."""
import math, os, sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import ops
torch.manual_seed(0)
B, H, S = 8, 24, 4608
for kind in ("random", "zeros", "random"):
    mk = (lambda: torch.randn(B, H, S, 128, device="cuda").bfloat16()) if kind == "random" else (lambda: torch.zeros(B, H, S, 128, device="cuda", dtype=torch.bfloat16))
    q, k, v = mk(), mk(), mk()
    vt = v.transpose(-1, -2).contiguous()
    O = torch.empty(B, S, H * 128, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B, H, S, device="cuda")
    fn = lambda: ops.attn_fwd(q, k, vt, O, lse, B, H, S, S, H * 128, S * H * 128, 1 / math.sqrt(128))
    for _ in range(20): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
    ms = sorted(ts)[2]
    print(f"MGX_ATTN_PP={os.environ.get('MGX_ATTN_PP', 'default')} {kind}: {ms:.3f} ms {4.0 * B * H * S * S * 128 / ms / 1e9:.0f} TFLOP/s", flush=True)
