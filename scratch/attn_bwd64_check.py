"""Timing of mgx_attn_bwd (prep + dK/dV + dQ) at B = 8, H = 24, S = 4608; `MGX_ATTN_W64=0` runs the 8-wave kernels."""
import math, os, sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import ops
print("MGX_ATTN_W64 =", os.environ.get("MGX_ATTN_W64", "1 (default)"))
B, H, S = 8, 24, 4608
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, 128, device="cuda").bfloat16() for _ in range(3))
tr = lambda t: t.transpose(-1, -2).contiguous()
vt, qt, kt = tr(v), tr(q), tr(k)
O = torch.empty(B, S, H * 128, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B, H, S, device="cuda")
sc = 1 / math.sqrt(128)
ops.attn_fwd(q, k, vt, O, lse, B, H, S, S, H * 128, S * H * 128, sc)
do = torch.randn_like(O); dQ, dK, dV = (torch.empty_like(q) for _ in range(3))
delta = torch.empty(B, H, S, device="cuda"); dOt = torch.zeros(B, H, 128, S, device="cuda", dtype=torch.bfloat16)
def run(): ops.attn_bwd(q, k, v, qt, kt, O, do, lse, delta, dOt, dQ, dK, dV, B, H, S, S, H * 128, S * H * 128, sc)
run(); torch.cuda.synchronize()
for rep in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5): run()
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"rep {rep}: {ms:.3f} ms  {10.0 * B * H * S * S * 128 / ms / 1e9:.0f} TFLOP/s (algorithmic 10 S^2 d)", flush=True)
