"""PMC / kernel-trace target (round 4, second half): a few launches of the rollout's new kernels at the FLUX shape B = 8 --
mgx_attn_fwd_log2 (attn_fwd64_kernel<true>), mgx_linear_qk_norm_rope on the pair table (gemm_pp_kernel<6, false, 0>) and
mgx_linear_bf16_t (gemm_pp_kernel<0, false, 0>, role-swapped value projection)."""
import math, sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import ops
torch.manual_seed(0)
B, rows, H, K, S = 8, 4608, 24, 3072, 4608
d, tokens = H * 128, B * rows
C = 1.4426950408889634 / math.sqrt(128)
q, k, v = (torch.randn(B, H, S, 128, device="cuda").bfloat16() for _ in range(3))
q2 = (q.float() * C).bfloat16(); vt = v.transpose(-1, -2).contiguous()
O = torch.empty(B, S, H * 128, device="cuda", dtype=torch.bfloat16)
X = (torch.randn(tokens, K, device="cuda") * 0.7).bfloat16()
W = (torch.randn(3 * d, K, device="cuda") * 0.02).bfloat16()
bias = torch.zeros(3 * d, device="cuda", dtype=torch.bfloat16)
wq = torch.ones(128, device="cuda"); wk = torch.ones(128, device="cuda")
cos = torch.rand(S, 64, device="cuda").repeat_interleave(2, dim=1).contiguous(); sin = torch.rand(S, 64, device="cuda").repeat_interleave(2, dim=1).contiguous()
pairs = ops.rope_pair_table(cos, sin)
Q = torch.empty(B, H, S, 128, device="cuda", dtype=torch.bfloat16); Kt = torch.empty_like(Q)
Vt = torch.empty(B, H, 128, S, device="cuda", dtype=torch.bfloat16)
ops.GEMM_STREAM_K = False
for _ in range(4):
    ops.attn_fwd_log2(q2, k, vt, O, None, B, H, S, S, H * 128, S * H * 128)
    assert ops.linear_qk_norm_rope(X, W[:2 * d], bias[:2 * d], wq, wk, cos, sin, Q, Kt, B, H, S, rows, 0, K, pairs=pairs)
    assert ops.linear_t(X, W[2 * d:], bias[2 * d:], Vt, tokens, d, K, S, rows, d * S)
torch.cuda.synchronize()
