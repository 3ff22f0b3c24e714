import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_hip_mmdit import build_pair, small_cfg, make_inputs
from mixgrpo_amd import flux_backward as FB
def run(keep):
    FB.KEEP_ACTS = keep
    ocfg, P, m = build_pair(small_cfg(2, 2))
    x, ehs, pooled, ids, tids, t, gd = make_inputs(2, 6, 10, 24, seed=3)
    R = torch.randn(2, 60, 64, generator=torch.Generator().manual_seed(9)).cuda()
    m.train()
    out = m(x.cuda(), ehs.cuda(), t.cuda(), gd.cuda(), tids.cuda(), pooled.cuda(), ids.cuda())[0]
    (out.float() * R).sum().backward()
    return m, m.store.g32.clone()
m, a = run(True); _, b = run(True); _, c = run(False); _, d = run(False)
print("keep vs keep equal", torch.equal(a, b), "full vs full", torch.equal(c, d), "keep vs full", torch.equal(a, c))
for k in m.store.index:
    ga, gc = m.store.view(a, k), m.store.view(c, k)
    if not torch.equal(ga, gc):
        print(k, ((ga - gc).norm() / (gc.norm() + 1e-12)).item(), (ga != gc).float().mean().item())
