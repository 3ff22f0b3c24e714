"""Correctness (vs an fp32 torch reference) and timing of mgx_attn_fwd at shapes the 64-query-wave kernel takes.
`MGX_ATTN_W64=0 python scratch/attn_fwd64_check.py` runs the 8-wave kernel on the same inputs."""
import math, os, sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import ops

def rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()

def one(B, H, S, seed, ldo_mult=1, spike=False):
    g = torch.Generator(device="cuda").manual_seed(seed)
    q = torch.randn(B, H, S, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(B, H, S, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(B, H, S, 128, device="cuda", generator=g).bfloat16()
    if spike:
        k[:, :, 200] = (8 * q[:, :, 70].float()).bfloat16()
        k[:, :, S - 3] = (6 * q[:, :, 100].float()).bfloat16()
    vt = v.transpose(-1, -2).contiguous()
    ldo = H * 128 * ldo_mult
    O = torch.zeros(B, S, ldo, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, H, S, device="cuda")
    ops.attn_fwd(q, k, vt, O, lse, B, H, S, S, ldo, S * ldo, 1 / math.sqrt(128))
    torch.cuda.synchronize()
    s = (q.float() @ k.float().transpose(-1, -2)) / math.sqrt(128)
    ref = (torch.softmax(s, -1) @ v.float()).transpose(1, 2).reshape(B, S, H * 128)
    e = rel(O[:, :, :H * 128], ref)
    le = (lse - torch.logsumexp(s, -1)).abs().max().item()
    untouched = O[:, :, H * 128:].abs().max().item() if ldo_mult > 1 else 0.0
    print(f"B={B} H={H} S={S} ldo={ldo} spike={spike}: rel {e:.3e}  lse err {le:.2e}  finite {bool(torch.isfinite(O.float()).all())} beyond-cols {untouched}", flush=True)
    assert e < 6e-3 and le < 2e-3 and untouched == 0.0

print("MGX_ATTN_W64 =", os.environ.get("MGX_ATTN_W64", "1 (default)"))
one(1, 1, 256, 0); one(1, 2, 512, 1); one(2, 3, 768, 2, ldo_mult=5); one(1, 2, 1024, 3, spike=True); one(1, 24, 1536, 4)
B, H, S = 8, 24, 4608
torch.manual_seed(0)
q = torch.randn(B, H, S, 128, device="cuda").bfloat16(); k = torch.randn(B, H, S, 128, device="cuda").bfloat16()
v = torch.randn(B, H, S, 128, device="cuda").bfloat16(); vt = v.transpose(-1, -2).contiguous()
O = torch.empty(B, S, H * 128, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B, H, S, device="cuda")
def run(): ops.attn_fwd(q, k, vt, O, lse, B, H, S, S, H * 128, S * H * 128, 1 / math.sqrt(128))
run(); torch.cuda.synchronize()
rows = torch.randint(0, S, (64,), generator=torch.Generator().manual_seed(1)).cuda()
s = (q[:, :, rows].float() @ k.float().transpose(-1, -2)) / math.sqrt(128)
ref = torch.softmax(s, -1) @ v.float()
got = O.view(B, S, H, 128).permute(0, 2, 1, 3)[:, :, rows]
print(f"full size sampled rows: rel {rel(got, ref):.3e}  lse err {(lse[:, :, rows] - torch.logsumexp(s, -1)).abs().max().item():.2e}", flush=True)
for rep in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(10): run()
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"rep {rep}: {ms:.3f} ms  {4.0 * B * H * S * S * 128 / ms / 1e9:.0f} TFLOP/s", flush=True)
