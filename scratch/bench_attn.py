import sys, math, torch
sys.path.insert(0, ".")
from mixgrpo_amd import ops
torch.manual_seed(0)
B, H, S = 8, 24, 4608
Sp = S
q = torch.randn(B, H, S, 128, device="cuda").bfloat16(); k = torch.randn(B, H, S, 128, device="cuda").bfloat16()
v = torch.randn(B, H, S, 128, device="cuda").bfloat16()
vt = v.transpose(-1, -2).contiguous(); qt = q.transpose(-1, -2).contiguous(); kt = k.transpose(-1, -2).contiguous()
O = torch.empty(B, S, H * 128, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B, H, S, device="cuda")
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = t(lambda: ops.attn_fwd(q, k, vt, O, lse, B, H, S, Sp, H * 128, S * H * 128, 1 / math.sqrt(128)))
fl = 4.0 * B * H * S * S * 128
print(f"attn fwd: {ms:.2f} ms  {fl/ms/1e9:.0f} TFLOP/s")
do = torch.randn_like(O); dQ, dK, dV = (torch.empty_like(q) for _ in range(3))
delta = torch.empty(B, H, S, device="cuda"); dOt = torch.zeros(B, H, 128, Sp, device="cuda", dtype=torch.bfloat16)
ms = t(lambda: ops.attn_bwd(q, k, v, qt, kt, O, do, lse, delta, dOt, dQ, dK, dV, B, H, S, Sp, H * 128, S * H * 128, 1 / math.sqrt(128)), 3)
print(f"attn bwd: {ms:.2f} ms  {2.5*fl/ms/1e9:.0f} TFLOP/s (algorithmic 10*S^2*d)")
qq = q.transpose(1, 2).contiguous()
ms = t(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v))
print(f"torch sdpa fwd: {ms:.2f} ms {fl/ms/1e9:.0f} TFLOP/s")
