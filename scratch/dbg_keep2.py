import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_hip_mmdit import build_pair, small_cfg, make_inputs
from mixgrpo_amd import flux_backward as FB
FB.KEEP_ACTS = True
ocfg, P, m = build_pair(small_cfg(2, 2))
x, ehs, pooled, ids, tids, t, gd = make_inputs(2, 6, 10, 24, seed=3)
m.train()
out = m(x.cuda(), ehs.cuda(), t.cuda(), gd.cuda(), tids.cuda(), pooled.cuda(), ids.cuda())[0]
fn = out.grad_fn
w = next(iter(m._work.values())); tr = w.train
sv = None
# reach into the autograd ctx
ctx = fn
saved = ctx.saved; st, cos, sin, mods = saved["st"], saved["cos"], saved["sin"], saved["mods"]
def snap(save):
    d = {k: v.clone() for k, v in save.items()}
    d.update(hid=w.hid.clone(), O=w.O.clone(), lse=w.lse.clone(), Q=w.Q.clone(), K=w.K.clone(), qkv=w.qkv.clone(), cat=w.cat.clone())
    return d
for blk, kind in ((1, "double"), (2, "single")):
    kept = tr.keep[blk]
    with torch.no_grad():
        w.X.copy_(tr.block_in[blk])
        if kind == "double":
            m._double_block(blk, w, st, cos, sin, save=tr.save, mods_in=mods[blk])
        else:
            m._single_block(blk - 2, w, st, cos, sin, save=tr.save, mod_in=mods[blk])
        a = snap(tr.save)
        for v in tr.save.values(): v.zero_() if v.dtype != torch.float32 else None
        w.X.copy_(tr.block_in[blk])
        save = dict(tr.save, **{k: kept[k] for k in kept if k in ("y_attn", "y_ff", "x_mid")})
        if kind == "double":
            m._double_block(blk, w, st, cos, sin, save=save, mods_in=mods[blk], keep=kept, replay=True)
        else:
            m._single_block(blk - 2, w, st, cos, sin, save=save, mod_in=mods[blk], keep=kept, replay=True)
        b = snap(save)
    for k in a:
        if kind == "single" and k in ("nrm2", "y_ff", "x_mid", "O"): continue
        eq = torch.equal(a[k], b[k])
        print(kind, k, "equal" if eq else f"DIFF {(a[k].float()-b[k].float()).abs().max().item():.4g} frac {(a[k]!=b[k]).float().mean().item():.4f}")
