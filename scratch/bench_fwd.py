import sys, time, torch
sys.path.insert(0, ".")
from mixgrpo_amd.flux import FluxConfig, FluxTransformer2DModel
torch.manual_seed(0)
t0 = time.time()
m = FluxTransformer2DModel(FluxConfig(), device="cuda").init_synthetic(seed=0)
torch.cuda.synchronize(); print("init", time.time() - t0, "s; params", m.store.numel / 1e9, "B; mem", torch.cuda.memory_allocated() / 2**30, "GiB")
m.eval()
def run(B, iters=3):
    N, L = 4096, 512
    x = torch.randn(B, N, 64, device="cuda"); ehs = (0.1 * torch.randn(B, L, 4096, device="cuda")).bfloat16()
    pooled = torch.randn(B, 768, device="cuda").bfloat16()
    ids = torch.zeros(64, 64, 3); ids[..., 1] += torch.arange(64)[:, None]; ids[..., 2] += torch.arange(64)[None]
    ids = ids.reshape(N, 3).cuda().bfloat16(); tids = torch.zeros(L, 3, device="cuda")
    t = torch.full([B], 0.954, device="cuda"); gd = torch.tensor([3.5], device="cuda").bfloat16()
    out = m(x, ehs, t, gd, tids, pooled, ids)[0]
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(iters): out = m(x, ehs, t, gd, tids, pooled, ids)[0]
    torch.cuda.synchronize(); dt = (time.time() - t0) / iters
    print(f"B={B}: {dt*1e3:.1f} ms/fwd  {74.38*B/dt/1e3:.1f} TFLOP/s... out finite {torch.isfinite(out.float()).all().item()} absmean {out.float().abs().mean().item():.4f}")
run(1); run(4); run(8)
