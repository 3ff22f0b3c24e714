"""Rollout vs replay: is the replayed log-prob equal to the rollout's on unchanged weights?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mixgrpo_amd import train_grpo_flux as TG
from mixgrpo_amd.flux import FluxConfig, FluxTransformer2DModel
from mixgrpo_amd.optim import ConstantWithWarmup, FusedAdamW
dev = torch.device("cuda", 0)
kw = dict(num_layers=2, num_single_layers=2, attention_head_dim=128, num_attention_heads=4, joint_attention_dim=64,
          pooled_projection_dim=32)
g = torch.Generator().manual_seed(0)
m = FluxTransformer2DModel(FluxConfig(**kw), device=dev).init_synthetic(seed=5, std=0.05, bias_std=0.02)
B, N, L = 4, 48, 16
xs = torch.randn(B, N, 64, generator=g).to(dev)
ehs = torch.randn(B, L, 64, generator=g).bfloat16().to(dev)
pooled = torch.randn(B, 32, generator=g).bfloat16().to(dev)
ids = torch.zeros(6, 8, 3); ids[..., 1] += torch.arange(6)[:, None]; ids[..., 2] += torch.arange(8)[None]
ids = ids.reshape(N, 3).to(dev)
t = torch.tensor([0.954, 0.5, 0.954, 0.2]).to(dev); gd = torch.tensor([3.5]).bfloat16().to(dev)
txt = torch.zeros(L, 3, device=dev)
m.eval()
with torch.autocast("cuda", torch.bfloat16):
    o_eval = m(xs, ehs, t, gd, txt, pooled, ids)[0].clone()
    o_eval1 = torch.cat([m(xs[i:i+1], ehs[i:i+1], t[i:i+1], gd, txt, pooled[i:i+1], ids)[0].clone() for i in range(B)])
m.train()
with torch.autocast("cuda", torch.bfloat16):
    o_train = m(xs, ehs, t, gd, txt, pooled, ids)[0]
print("train requires_grad", o_train.requires_grad)
print("eval(B=4) vs eval(B=1 x4): max abs diff", (o_eval.float() - o_eval1.float()).abs().max().item(), "equal", torch.equal(o_eval, o_eval1))
print("eval vs train: max abs diff", (o_eval.float() - o_train.float()).abs().max().item(), "equal", torch.equal(o_eval, o_train.detach()),
      "|o| max", o_eval.float().abs().max().item())
# second train forward (workspace reuse)
with torch.autocast("cuda", torch.bfloat16):
    o_train2 = m(xs, ehs, t, gd, txt, pooled, ids)[0]
print("train vs train2 equal", torch.equal(o_train.detach(), o_train2.detach()))

opt = FusedAdamW(m, lr=0.0)
args = TG.default_args(h=48, w=64, sampling_steps=6, num_generations=4, gradient_accumulation_steps=2)
loader = iter([(ehs[:1], pooled[:1], torch.zeros(1, 3, device=dev), ["p"])])
def reward(lat, cap):
    r = torch.tensor([0.1, 0.4, 0.2, 0.9]); return r, {"Synthetic": r}
trace = {}
res = TG.train_one_step(args, dev, m, None, reward, opt, ConstantWithWarmup(opt, 0), loader, None, 1.0, [1, 2], 0,
                        {"Synthetic": 1.0}, trace=trace)
print("res", res)
lp = trace["log_probs"]
print("rollout log_probs", lp.tolist())
for pairs, nl in trace["new_log_probs"]:
    old = torch.stack([lp[i, tt] for i, tt in pairs])
    print("pairs", pairs, "old", old.tolist(), "new", nl.tolist(), "diff", (nl - old).tolist())
