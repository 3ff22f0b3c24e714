import math, os, sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import ops
def rel(a, b): return ((a.float() - b.float()).norm() / b.float().norm()).item()
for (B, H, S, lm) in [(1, 1, 256, 1), (1, 2, 256, 1), (1, 1, 512, 1), (2, 3, 768, 5), (1, 6, 1536, 1)]:
    g = torch.Generator(device="cuda").manual_seed(S + 7 * H)
    q, k, v = (torch.randn(B, H, S, 128, device="cuda", generator=g).bfloat16() for _ in range(3))
    ldo = H * 128 * lm
    do = torch.randn(B, S, ldo, device="cuda", generator=g).bfloat16()
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    s = (qf @ kf.transpose(-1, -2)) / math.sqrt(128)
    o_ref = (torch.softmax(s, -1) @ vf).transpose(1, 2).reshape(B, S, H * 128)
    o_ref.backward(do[:, :, :H * 128].float())
    tr = lambda t: t.transpose(-1, -2).contiguous()
    vt, qt, kt = tr(v), tr(q), tr(k)
    for w64 in ("1", "0"):
        os.environ["MGX_ATTN_W64"] = w64
        O = torch.zeros(B, S, ldo, device="cuda", dtype=torch.bfloat16)
        lse = torch.empty(B, H, S, device="cuda")
        ops.attn_fwd(q, k, vt, O, lse, B, H, S, S, ldo, S * ldo, 1 / math.sqrt(128))
        dQ, dK, dV = (torch.full_like(q, float("nan")) for _ in range(3))
        delta = torch.empty(B, H, S, device="cuda")
        dOt = torch.zeros(B, H, 128, S, device="cuda", dtype=torch.bfloat16)
        ops.attn_bwd(q, k, v, qt, kt, O, do, lse, delta, dOt, dQ, dK, dV, B, H, S, S, ldo, S * ldo, 1 / math.sqrt(128))
        torch.cuda.synchronize()
        print(f"B={B} H={H} S={S} ldo={ldo} w64={w64}: dV {rel(dV, vf.grad):.3e} dK {rel(dK, kf.grad):.3e} dQ {rel(dQ, qf.grad):.3e} nan {bool(torch.isnan(dK.float()).any())} {bool(torch.isnan(dV.float()).any())}", flush=True)
        if w64 == "1":
            # per 32-key chain / per wave error map for dV
            e = ((dV.float() - vf.grad) ** 2).sum(-1).sqrt()[0, 0] / vf.grad[0, 0].norm(dim=-1)
            print("   dV row-err per 32-key group:", [f"{x:.2e}" for x in e.view(-1, 32).mean(1).tolist()[:16]])
            e = ((dK.float() - kf.grad) ** 2).sum(-1).sqrt()[0, 0] / kf.grad[0, 0].norm(dim=-1)
            print("   dK row-err per 32-key group:", [f"{x:.2e}" for x in e.view(-1, 32).mean(1).tolist()[:16]])
