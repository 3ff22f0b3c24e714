import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_hip_mmdit import build_pair, small_cfg, make_inputs
from mixgrpo_amd import flux_backward as FB
FB.KEEP_ACTS = True
ocfg, P, m = build_pair(small_cfg(2, 2))
x, ehs, pooled, ids, tids, t, gd = make_inputs(2, 6, 10, 24, seed=3)
m.train()
out = m(x.cuda(), ehs.cuda(), t.cuda(), gd.cuda(), tids.cuda(), pooled.cuda(), ids.cuda())[0]
ctx = out.grad_fn
w = next(iter(m._work.values())); tr = w.train
saved = ctx.saved; st, cos, sin, mods = saved["st"], saved["cos"], saved["sin"], saved["mods"]
def snap():
    return dict(Q=w.Q.clone(), K=w.K.clone(), Vt=w.Vt.clone(), O=w.O.clone(), qkv=w.qkv.clone(), nrm=w.nrm.clone())
blk = 1
with torch.no_grad():
    res = []
    for mode in ("fwd", "fwd", "rec", "rec"):
        w.X.copy_(tr.block_in[blk])
        if mode == "fwd":
            m._double_block(blk, w, st, cos, sin, keep=tr.keep[blk])
            s = snap(); s["lse"] = tr.keep[blk]["lse"].clone()
        else:
            m._double_block(blk, w, st, cos, sin, save=tr.save, mods_in=mods[blk])
            s = snap(); s["lse"] = w.lse.clone(); s["nrm"] = tr.save["nrm1"].clone()
        res.append(s)
for (i, j) in ((0, 1), (2, 3), (0, 2)):
    for k in res[0]:
        a, b = res[i][k], res[j][k]
        if not torch.equal(a, b):
            idx = (a != b).nonzero()
            print(i, j, k, "DIFF n=", idx.shape[0], "first", idx[:4].tolist(), "last", idx[-2:].tolist())
        else:
            print(i, j, k, "equal")
