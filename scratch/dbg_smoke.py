import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mixgrpo_amd import train_grpo_flux as TG
from mixgrpo_amd.flux import FluxConfig, FluxTransformer2DModel
from mixgrpo_amd.optim import ConstantWithWarmup, FusedAdamW
from oracle import mmdit as OM
dev = torch.device("cuda", 0)
kw = dict(num_layers=1, num_single_layers=1, attention_head_dim=128, num_attention_heads=4, joint_attention_dim=64,
          pooled_projection_dim=32)
for variant in ("eval_first", "no_eval", "synthetic"):
    g = torch.Generator().manual_seed(0)
    ocfg = OM.FluxConfig(**kw)
    P = OM.init_params(ocfg, seed=1, std=0.05, bias_std=0.02)
    m = FluxTransformer2DModel(FluxConfig(**kw), device=dev)
    if variant == "synthetic":
        m.init_synthetic(seed=5, std=0.05, bias_std=0.02)
    else:
        m.load_state_dict({k: t.to(dev) for k, t in P.items()})
    B, N, L = 2, 48, 16
    xs = torch.randn(B, N, 64, generator=g)
    ehs = torch.randn(B, L, 64, generator=g).bfloat16()
    pooled = torch.randn(B, 32, generator=g).bfloat16()
    ids = torch.zeros(6, 8, 3); ids[..., 1] += torch.arange(6)[:, None]; ids[..., 2] += torch.arange(8)[None]
    ids = ids.reshape(N, 3)
    t = torch.tensor([0.954, 0.5]); gd = torch.tensor([3.5]).bfloat16()
    if variant == "eval_first":
        m.eval()
        out = m(xs.to(dev), ehs.to(dev), t.to(dev), gd.to(dev), torch.zeros(L, 3, device=dev), pooled.to(dev), ids.to(dev))[0]
    opt = FusedAdamW(m, lr=1e-4)
    args = TG.default_args(h=48, w=64, sampling_steps=6, num_generations=4, gradient_accumulation_steps=2)
    loader = iter([(ehs[:1].to(dev), pooled[:1].to(dev), torch.zeros(1, 3, device=dev), ["smoke"])])
    def reward(lat, cap):
        r = torch.tensor([0.1, 0.4, 0.2, 0.9]); return r, {"Synthetic": r}
    w_before = m.store.w32.clone()
    trace = {}
    res = TG.train_one_step(args, dev, m, None, reward, opt, ConstantWithWarmup(opt, 0), loader, None, 1.0, [1, 2], 0,
                            {"Synthetic": 1.0}, trace=trace) if "trace" in TG.train_one_step.__code__.co_varnames else \
          TG.train_one_step(args, dev, m, None, reward, opt, ConstantWithWarmup(opt, 0), loader, None, 1.0, [1, 2], 0, {"Synthetic": 1.0})
    torch.cuda.synchronize()
    d = (m.store.w32 - w_before).abs()
    print(variant, "res", res, "wdiff max", d.max().item(), "nnz", (d > 0).sum().item(), "training", m.training,
          "flat req", m.flat_param.requires_grad, flush=True)
    if "advantages" in trace: print("  adv", trace["advantages"].tolist(), "g_logp", [x.tolist() for x in trace.get("g_logp", [])])
