"""Tile-order band (MGX_GEMM_BAND, read once per process) for the rollout's split projections: the fused q | k launch
(36864 x 6144 x 3072, norm epilogue on the pair table) and the role-swapped value projection (3072 x 36864 x 3072)."""
import json, os, sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import ops
torch.manual_seed(0)
B, rows, H, K, S = 8, 4608, 24, 3072, 4608
d, tokens = H * 128, B * rows
X = (torch.randn(tokens, K, device="cuda") * 0.7).bfloat16()
W = (torch.randn(3 * d, K, device="cuda") * 0.02).bfloat16()
bias = torch.zeros(3 * d, device="cuda", dtype=torch.bfloat16)
wq = torch.ones(128, device="cuda"); wk = torch.ones(128, device="cuda")
cos = torch.rand(S, 64, device="cuda").repeat_interleave(2, dim=1).contiguous(); sin = torch.rand(S, 64, device="cuda").repeat_interleave(2, dim=1).contiguous()
pairs = ops.rope_pair_table(cos, sin)
Q = torch.empty(B, H, S, 128, device="cuda", dtype=torch.bfloat16); Kt = torch.empty_like(Q)
Vt = torch.empty(B, H, 128, S, device="cuda", dtype=torch.bfloat16)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
qk = lambda: ops.linear_qk_norm_rope(X, W[:2 * d], bias[:2 * d], wq, wk, cos, sin, Q, Kt, B, H, S, rows, 0, K, pairs=pairs)
vt = lambda: ops.linear_t(X, W[2 * d:], bias[2 * d:], Vt, tokens, d, K, S, rows, d * S)
a = min(t(qk) for _ in range(3)); b = min(t(vt) for _ in range(3))
print(json.dumps({"band": os.environ.get("MGX_GEMM_BAND", "default"), "qk_ms": round(a, 4), "vt_ms": round(b, 4)}), flush=True)
