import sys, torch
sys.path.insert(0, ".")
from mixgrpo_amd import ops
from mixgrpo_amd.ops import Rows
for (M, N) in [(27648, 3072), (27648, 12288), (1000, 72), (300, 136), (4608, 64)]:
    x = torch.randn(M, N, device="cuda").bfloat16()
    Mp = (M + 63) // 64 * 64
    out = torch.full((N, Mp), 7.0, device="cuda", dtype=torch.bfloat16)
    cs = torch.ones(N, device="cuda")
    ops.transpose(Rows.of(x), N, out, Mp, colsum_out=cs, colsum_beta=1.0)
    ok = torch.equal(out[:, :M], x.t()) and (out[:, M:] == 0).all().item()
    ref = 1.0 + x.float().sum(0)
    print(M, N, "ok" if ok else "MISMATCH", "colsum err", (cs - ref).abs().max().item())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.transpose(Rows.of(x), N, out, Mp)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"   {ms*1e3:.1f} us  {4*M*N/ms/1e6:.0f} GB/s")
# row-batched input
B, S, L, d = 2, 200, 40, 128
X = torch.randn(B, S, d, device="cuda").bfloat16()
r = Rows(X[0, L:], B * (S - L), d, S - L, S * d)
Mp = (r.M + 63) // 64 * 64
out = torch.empty(d, Mp, device="cuda", dtype=torch.bfloat16)
ops.transpose(r, d, out, Mp)
print("batched", torch.equal(out[:, :r.M], X[:, L:].reshape(-1, d).t()))
