import sys, math, torch
sys.path.insert(0, ".")
from mixgrpo_amd import ops
torch.manual_seed(0)
B, H, S = 8, 24, 4608
q = torch.randn(B, H, S, 128, device="cuda").bfloat16(); k = torch.randn(B, H, S, 128, device="cuda").bfloat16()
v = torch.randn(B, H, S, 128, device="cuda").bfloat16()
vt = v.transpose(-1, -2).contiguous(); qt = q.transpose(-1, -2).contiguous(); kt = k.transpose(-1, -2).contiguous()
O = torch.empty(B, S, H * 128, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B, H, S, device="cuda")
for _ in range(3): ops.attn_fwd(q, k, vt, O, lse, B, H, S, S, H * 128, S * H * 128, 1 / math.sqrt(128))
do = torch.randn_like(O); dQ, dK, dV = (torch.empty_like(q) for _ in range(3))
delta = torch.empty(B, H, S, device="cuda"); dOt = torch.zeros(B, H, 128, S, device="cuda", dtype=torch.bfloat16)
for _ in range(2): ops.attn_bwd(q, k, v, qt, kt, O, do, lse, delta, dOt, dQ, dK, dV, B, H, S, S, H * 128, S * H * 128, 1 / math.sqrt(128))
torch.cuda.synchronize()
