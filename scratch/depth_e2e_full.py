"""One-off evidence run: ONE GRPO train step at FLUX.1-dev's FULL depth -- 19 + 38 blocks, 11.9 B
parameters, full width -- HIP engine against the oracle (oracle/trainer.py driving oracle/mmdit.py through torch autograd +
torch.optim.AdamW), to record the replayed log-prob difference AFTER the first optimizer update at full depth.

The configuration is tests/test_hip_e2e_mmdit.py::test_train_step_at_depth_vs_oracle's (two samples, one per optimizer step,
four sampler steps, window [1, 2], lr 1e-5, 64 x 64 image = 16 image tokens + 8 text tokens).  Two things differ, both forced
by size: (1) the oracle's MMDiT runs with its parameters ON THE GPU (the same device-agnostic fp32 torch code; its AdamW state
for 11.9 B parameters is 95 GB and the host pass would take tens of minutes), inputs moved over and the output moved back per
call, everything else of the trainer oracle on the host as always; (2) the two sides run ONE AFTER THE OTHER (oracle: 190 GB
of HBM, HIP engine: 214 GB), from the same host copy of the weights.  Writes gpurun_out/r04_depth_e2e_full.json."""
import copy, json, os, sys, time
from argparse import Namespace
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import mmdit as OM
from oracle import trainer as OT

t00 = time.time()
dev = torch.device("cuda", 0)
KW = dict(num_layers=int(os.environ.get("ND", "19")), num_single_layers=int(os.environ.get("NS", "38")))
hw, std, lr, rewards, window = 64, 0.02, 1e-5, [0.2, 0.8], [1, 2]
a = Namespace(w=hw, h=hw, t=1, sampling_steps=4, shift=3.0, init_same_noise=True, training_strategy="part",
              output_dir="/tmp/x", experiment_name="t", reward_model="toy", multi_reward_mix="advantage_aggr",
              use_group=True, num_generations=2, trimmed_ratio=0.0, advantage_rerange_strategy="null", clip_range=1e-4,
              adv_clip_max=5.0, kl_coeff=0.0, gradient_accumulation_steps=1, frozen_init_timesteps=-1,
              timestep_fraction=1.0, dpm_algorithm_type="null", dpm_apply_strategy="post", dpm_post_compress_ratio=0.4,
              dpm_solver_order=2, dpm_solver_type="midpoint", sample_strategy="progressive", flow_grpo_sampling=True,
              eta=0.7, drop_last_sample=False, rollout_batch=0, train_microbatch=0)
G, T = a.num_generations, a.sampling_steps
lh = lw = hw // 8
N = (lh // 2) * (lw // 2)
g = torch.Generator().manual_seed(11)
inj = {"x_T": torch.randn(1, 16, lh, lw, generator=g).bfloat16(),
       "steps": [torch.randn(G, N, 64, generator=g).bfloat16() for _ in range(T)]}
ocfg = OM.FluxConfig(**KW)
ehs = (0.5 * torch.randn(1, 8, ocfg.joint_attention_dim, generator=g)).bfloat16()
pooled = torch.randn(1, ocfg.pooled_projection_dim, generator=g).bfloat16()
text_ids = torch.zeros(1, 3)
weights = {"A": 1.0}

# weights: drawn on the device tensor by tensor (the distributions of OM.init_params(std, bias_std = 0.02)), host copy kept
gd = torch.Generator(device=dev).manual_seed(3)
P_cpu = {}
for k, shp in OM.param_shapes(ocfg).items():
    if k.endswith(".bias"):
        t = torch.randn(shp, generator=gd, device=dev) * 0.02
    elif len(shp) == 1:
        t = 1.0 + torch.randn(shp, generator=gd, device=dev) * 0.02
    else:
        t = torch.randn(shp, generator=gd, device=dev) * std
    P_cpu[k] = t.cpu()
    del t
print(f"weights: {sum(v.numel() for v in P_cpu.values()) / 1e9:.2f} B parameters, {time.time() - t00:.0f} s", flush=True)


class _Sched:
    def step(self):
        pass


class DevOracle(torch.nn.Module):
    """helpers.oracle_flux with the parameters (and the MMDiT restatement's arithmetic) on the GPU."""

    def __init__(self):
        super().__init__()
        self.cfg, self.names = ocfg, list(P_cpu)
        self.params = torch.nn.ParameterList([torch.nn.Parameter(P_cpu[k].to(dev)) for k in self.names])
        self.config = {"oracle": True}

    def forward(self, hidden_states, encoder_hidden_states, timestep, guidance, txt_ids, pooled_projections, img_ids,
                joint_attention_kwargs=None, return_dict=False):
        Pd = dict(zip(self.names, self.params))
        f = lambda x: x.float().to(dev)
        out = OM.forward(Pd, self.cfg, f(hidden_states), f(encoder_hidden_states), f(timestep), f(guidance), f(txt_ids),
                         f(pooled_projections), f(img_ids))
        return (out.to(torch.bfloat16).cpu(),)

    def clip_grad_norm_(self, max_norm):
        return torch.nn.utils.clip_grad_norm_(self.parameters(), max_norm).cpu()


def o_reward(i, latents):
    return [rewards[i]], {"A": [rewards[i]]}


def p_reward(latents, captions):
    n = latents.shape[0]
    return [rewards[i] for i in range(n)], {"A": [rewards[i] for i in range(n)]}


# ---- phase 1: the oracle
mo = DevOracle()
oo = torch.optim.AdamW(mo.parameters(), lr=lr, betas=(0.9, 0.999), weight_decay=1e-4, eps=1e-8, foreach=False)   # (no param-sized temporaries)
tro = {}
ro = OT.train_one_step(a, mo, oo, _Sched(), (ehs, pooled, text_ids, ["p"]), o_reward, weights, window, 1.0, trace=tro, injected=inj)
ro = [float(x) if x is not None and not isinstance(x, dict) else x for x in ro]
print(f"oracle step done, {time.time() - t00:.0f} s, HBM peak {torch.cuda.max_memory_allocated() / 2**30:.0f} GiB", flush=True)
del mo, oo
import gc
gc.collect()
torch.cuda.empty_cache()

# ---- phase 2: the HIP engine
from mixgrpo_amd import train_grpo_flux as TG
from mixgrpo_amd.flux import FluxConfig, FluxTransformer2DModel
from mixgrpo_amd.optim import FusedAdamW
mp = FluxTransformer2DModel(FluxConfig(**KW), device=dev)
mp.load_state_dict({k: t for k, t in P_cpu.items()})
po = FusedAdamW(mp, lr=lr, betas=(0.9, 0.999), weight_decay=1e-4, eps=1e-8)
ap = copy.copy(a)
ap.injected_noise = inj
trp = {}
rp = TG.train_one_step(ap, dev, mp, None, p_reward, po, _Sched(), iter([(ehs.to(dev), pooled.to(dev), text_ids.to(dev), ["p"])]),
                       None, 1.0, window, 0, weights, trace=trp)
torch.cuda.synchronize()
print(f"HIP step done, {time.time() - t00:.0f} s", flush=True)

# ---- the comparison of tests/test_hip_e2e_mmdit.py::_train_step_vs_oracle
lo, lp = tro["log_probs"], trp["log_probs"].cpu()
fin = torch.isfinite(lo)
new_p = {tuple(pr): v.cpu() for pairs, v in trp["new_log_probs"] for pr, v in zip(pairs, v)}
new_o = {(i, t): tro["new_log_probs"][i * len(window) + k] for i in range(G) for k, t in enumerate(window)}
shifts, diffs = [], []
for (i, t), v in new_p.items():
    vo = float(new_o[(i, t)])
    shifts.append(abs(vo - lo[i, t].item()))
    diffs.append(abs(v.item() - vo))
acc = a.gradient_accumulation_steps
first = [d for ((i, t), d) in zip(new_p, diffs) if i < acc]
second = [d for ((i, t), d) in zip(new_p, diffs) if i >= acc]
moved = [s for ((i, t), s) in zip(new_p, shifts) if i >= acc]
out = dict(blocks=[KW["num_layers"], KW["num_single_layers"]], rollout_logp_diff=(lp[fin] - lo[fin]).abs().max().item(),
           same_weights_replay_diff=max(first), max_after_update=max(second), max_shift_by_update=max(moved),
           after_update_over_shift=max(second) / max(moved) if max(moved) > 0 else None, loss=[rp[0], ro[0]],
           grad_norm=[rp[1], ro[1]], clip_frac=[rp[4], ro[4]], seconds=round(time.time() - t00),
           note="oracle MMDiT arithmetic on the GPU (fp32 torch), sides run one after the other; scratch/depth_e2e_full.py")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(os.path.join("gpurun_out", "r04_depth_e2e_full.json"), "w"), indent=1)
print(json.dumps(out, indent=1), flush=True)
