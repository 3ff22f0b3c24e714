"""PMC / kernel-trace target: a few launches of the attention kernels (bf16 forward, backward, fp8 quantiser + forward)
at the FLUX shape B=8, H=24, S=4608, and of the solver step at 512 images per call."""
import sys, math, torch
sys.path.insert(0, ".")
from mixgrpo_amd import ops
from mixgrpo_amd import sampling_utils as SU
torch.manual_seed(0)
B, H, S = 8, 24, 4608
q = torch.randn(B, H, S, 128, device="cuda").bfloat16(); k = torch.randn(B, H, S, 128, device="cuda").bfloat16()
v = torch.randn(B, H, S, 128, device="cuda").bfloat16()
vt = v.transpose(-1, -2).contiguous(); qt = q.transpose(-1, -2).contiguous(); kt = k.transpose(-1, -2).contiguous()
O = torch.empty(B, S, H * 128, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B, H, S, device="cuda")
sc = 1 / math.sqrt(128)
for _ in range(3): ops.attn_fwd(q, k, vt, O, lse, B, H, S, S, H * 128, S * H * 128, sc)
do = torch.randn_like(O); dQ, dK, dV = (torch.empty_like(q) for _ in range(3))
delta = torch.empty(B, H, S, device="cuda"); dOt = torch.zeros(B, H, 128, S, device="cuda", dtype=torch.bfloat16)
for _ in range(2): ops.attn_bwd(q, k, v, qt, kt, O, do, lse, delta, dOt, dQ, dK, dV, B, H, S, S, H * 128, S * H * 128, sc)
u8 = lambda *s: torch.empty(*s, dtype=torch.uint8, device="cuda")
Q8, K8, V8t, amax = u8(B, H, S, 128), u8(B, H, S, 128), u8(B, H, 128, S), torch.empty(3 * B * H, device="cuda")
for _ in range(3):
    ops.attn_fp8_quantize(q, k, vt, Q8, K8, V8t, amax, B, H, S, S)
    ops.attn_fwd_fp8(Q8, K8, V8t, amax, O, lse, B, H, S, S, H * 128, S * H * 128, sc)
# solver step (rollout form: x fp32, v bf16, noise bf16 -> x' fp32 + log-prob), 512 images of 4096 x 64 per call
n_img = 512
x = torch.randn(n_img, 4096, 64, device="cuda"); vv = torch.randn(n_img, 4096, 64, device="cuda").bfloat16()
nz = torch.randn(n_img, 4096, 64, device="cuda").bfloat16()
sig = SU.sd3_time_shift(3.0, torch.linspace(1, 0, 26))
for _ in range(3): SU.flow_grpo_step(vv, x, 0.7, sig, 2, None, noise=nz)
torch.cuda.synchronize()
