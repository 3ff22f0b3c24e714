"""Is the attention backward clock-limited?  Same kernels on random and on all-zero operands."""
import math, sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import ops
torch.manual_seed(0)
B, H, S = 8, 24, 4608
Sp = S
for kind in ("random", "zeros", "random"):
    mk = (lambda *s: torch.randn(*s, device="cuda").bfloat16()) if kind == "random" else (lambda *s: torch.zeros(*s, device="cuda", dtype=torch.bfloat16))
    q, k, v = mk(B, H, S, 128), mk(B, H, S, 128), mk(B, H, S, 128)
    vt = v.transpose(-1, -2).contiguous(); qt = q.transpose(-1, -2).contiguous(); kt = k.transpose(-1, -2).contiguous()
    O = torch.empty(B, S, H * 128, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B, H, S, device="cuda")
    ops.attn_fwd(q, k, vt, O, lse, B, H, S, Sp, H * 128, S * H * 128, 1 / math.sqrt(128))
    do = mk(B, S, H * 128); dQ, dK, dV = (torch.empty_like(q) for _ in range(3))
    delta = torch.empty(B, H, S, device="cuda"); dOt = torch.zeros(B, H, 128, Sp, device="cuda", dtype=torch.bfloat16)
    fn = lambda: ops.attn_bwd(q, k, v, qt, kt, O, do, lse, delta, dOt, dQ, dK, dV, B, H, S, Sp, H * 128, S * H * 128, 1 / math.sqrt(128))
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    ms = sorted(ts)[2]
    print(f"attn bwd {kind}: {ms:.3f} ms {10.0 * B * H * S * S * 128 / ms / 1e9:.0f} TFLOP/s (algorithmic 10 S^2 d)", flush=True)
