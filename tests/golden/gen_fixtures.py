#!/usr/bin/env python3
"""Generate golden vectors by running the *reference's own functions* in this container.

Run once, here (the reference lives at /root/reference and never travels):
    python tests/golden/gen_fixtures.py
Outputs (committed): tests/golden/*.safetensors, tests/golden/*.json.

Only inputs and expected outputs are stored; no reference source text is copied.  The
reference modules are loaded in-place with `importlib` (bytecode writing disabled).  The
absent third-party packages (diffusers, wandb, cv2, ...) are satisfied by inert stubs, as
recorded in SURVEY.md section 8c; `randn_tensor` is the only third-party symbol the sampler
actually executes and it is provided as `torch.randn` (that is what diffusers' helper does
for a CPU generator / no generator).
"""
import importlib.abc
import importlib.machinery
import importlib.util
import json
import os
import sys
import types
from argparse import Namespace

sys.dont_write_bytecode = True
import numpy as np
import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)
from toy_model import ToyTransformer  # noqa: E402

NOISE_LOG = []


def _randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    t = torch.randn(tuple(shape), generator=generator, device=device, dtype=dtype)
    NOISE_LOG.append(t.clone())
    return t


def install_light_stub():
    d = types.ModuleType("diffusers")
    du = types.ModuleType("diffusers.utils")
    dt = types.ModuleType("diffusers.utils.torch_utils")
    dt.randn_tensor = _randn_tensor
    d.utils = du
    du.torch_utils = dt
    sys.modules.update({"diffusers": d, "diffusers.utils": du, "diffusers.utils.torch_utils": dt})


def load_ref(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def dpm_args(**kw):
    base = dict(dpm_algorithm_type="null", dpm_apply_strategy="post", dpm_post_compress_ratio=0.4,
                dpm_solver_order=2, dpm_solver_type="midpoint", sample_strategy="progressive",
                shift=3.0, flow_grpo_sampling=True, eta=0.7, drop_last_sample=False)
    base.update(kw)
    return Namespace(**base)


# --------------------------------------------------------------------------- solver fixtures
def gen_solver(su):
    tensors, meta = {}, {"cases": []}
    # a1 sigma schedules
    sched = {}
    for shift, T in [(3.0, 8), (3.0, 25), (3.0, 50), (1.0, 10), (5.0, 16)]:
        s = su.sd3_time_shift(shift, torch.linspace(1, 0, T + 1))
        tensors[f"sigma/shift{shift}_T{T}"] = s.clone()
        sched[f"shift{shift}_T{T}"] = {"timesteps": [int(x * 1000) for x in s]}
    meta["schedules"] = sched

    B, N, C = 2, 6, 64
    g = torch.Generator().manual_seed(20250824)
    x = torch.randn(B, N, C, generator=g)
    v = torch.randn(B, N, C, generator=g).to(torch.bfloat16)
    tensors["in/x"] = x
    tensors["in/v"] = v

    # a3 flow_grpo_step: SDE, deterministic, replay, for several (T, index)
    for T, idxs in [(8, [0, 1, 3, 6, 7]), (25, [0, 1, 2, 3, 12, 22, 23])]:
        sig = su.sd3_time_shift(3.0, torch.linspace(1, 0, T + 1))
        for i in idxs:
            for det in (False, True):
                NOISE_LOG.clear()
                torch.manual_seed(1000 + 10 * i + int(det))
                out = su.flow_grpo_step(v, x, 0.7, sig, i, None, determistic=det)
                key = f"flow/T{T}_i{i}_det{int(det)}"
                tensors[key + "/noise"] = NOISE_LOG[0]
                for nm, t in zip(("prev", "x0", "logp", "mean", "std"), out):
                    tensors[f"{key}/{nm}"] = t.clone().contiguous()
                meta["cases"].append({"kind": "flow", "key": key, "T": T, "index": i, "det": det, "eta": 0.7})
                if not det:
                    # replay: log-prob of the stored next latent (training path, prev_sample given)
                    rp = su.flow_grpo_step(v, x, 0.7, sig, i, out[0].clone(), determistic=False)
                    tensors[f"{key}/replay_logp"] = rp[2].clone()

    # a4 dance_grpo_step
    sig = su.sd3_time_shift(3.0, torch.linspace(1, 0, 9))
    for i in (0, 2, 6):
        for sde in (False, True):
            torch.manual_seed(7 + i)
            st = torch.get_rng_state()
            out = su.dance_grpo_step(v, x, 0.3, sig, i, None, True, sde)
            torch.set_rng_state(st)
            noise = torch.randn_like(x)  # what randn_like drew inside (fp32, same generator state)
            key = f"dance/i{i}_sde{int(sde)}"
            tensors[key + "/noise"] = noise
            for nm, t in zip(("prev", "x0", "logp"), out):
                tensors[f"{key}/{nm}"] = t.clone().contiguous()
            meta["cases"].append({"kind": "dance", "key": key, "T": 8, "index": i, "sde": sde, "eta": 0.3})
            rp = su.dance_grpo_step(v, x, 0.3, sig, i, out[0].clone(), True, True)
            tensors[f"{key}/replay_logp_sde"] = rp[2].clone()

    # a5 dpm_step: multistep chains so that order 1/2/3 updates are all exercised
    for algo in ("dpmsolver++", "dpmsolver"):
        for order in (1, 2, 3):
            for stype in ("midpoint", "heun"):
                for sde in (False, True):
                    if algo == "dpmsolver" and order == 3:
                        # unreachable in the reference: SDE is asserted away (sampling_utils.py:630) and the
                        # ODE branch returns an unbound `prev_mean` (:639 -> UnboundLocalError)
                        continue
                    T = 8
                    sig = su.sd3_time_shift(3.0, torch.linspace(1, 0, T + 1))
                    a = dpm_args(dpm_algorithm_type=algo, dpm_solver_order=order, dpm_solver_type=stype)
                    st = su.DPMState(order=order)
                    key = f"dpm/{algo}_o{order}_{stype}_sde{int(sde)}"
                    xs = x.clone()
                    gg = torch.Generator().manual_seed(99)
                    for i in range(T):
                        vi = (v.float() * (1.0 - 0.07 * i) + 0.01 * i).to(torch.bfloat16)
                        noise = torch.randn(B, N, C, generator=gg) if sde else None
                        prev, x0, lp = su.dpm_step(a, vi, xs, i, sig[:-1], sig, dpm_state=st,
                                                   variance_noise=noise, sde_solver=sde)
                        tensors[f"{key}/s{i}/v"] = vi
                        if sde:
                            tensors[f"{key}/s{i}/noise"] = noise
                        tensors[f"{key}/s{i}/prev"] = prev.clone()
                        if i in (0, 3):
                            tensors[f"{key}/s{i}/x0"] = x0.clone()
                        tensors[f"{key}/s{i}/logp"] = lp.clone()
                        xs = prev
                    meta["cases"].append({"kind": "dpm", "key": key, "algo": algo, "order": order,
                                          "stype": stype, "sde": sde, "T": T})
    # dpm_step without state (training replay under strategy "all", train_grpo_flux.py:170-180)
    a = dpm_args(dpm_algorithm_type="dpmsolver++")
    sig = su.sd3_time_shift(3.0, torch.linspace(1, 0, 9))
    gg = torch.Generator().manual_seed(5)
    noise = torch.randn(B, N, C, generator=gg)
    prev, x0, lp = su.dpm_step(a, v, x, 3, sig[:-1], sig, dpm_state=None, variance_noise=noise, sde_solver=True)
    tensors["dpm/nostate/noise"] = noise
    tensors["dpm/nostate/prev"] = prev
    tensors["dpm/nostate/logp"] = lp
    # ... and its gradient w.r.t. the model output (the reference differentiates through prev_sample_mean, :376-383),
    # for both algorithm types, with a per-sample upstream gradient
    for algo in ("dpmsolver++", "dpmsolver"):
        a = dpm_args(dpm_algorithm_type=algo)
        vg = v.clone().requires_grad_(True)
        _, _, lpg = su.dpm_step(a, vg, x, 3, sig[:-1], sig, dpm_state=None, variance_noise=noise, sde_solver=True)
        up = torch.tensor([0.75, -1.5])
        (lpg * up).sum().backward()
        tensors[f"dpm/nostate_grad/{algo}/logp"] = lpg.detach().clone()
        tensors[f"dpm/nostate_grad/{algo}/upstream"] = up
        tensors[f"dpm/nostate_grad/{algo}/grad_v"] = vg.grad.clone()
    return tensors, meta


# --------------------------------------------------------------------------- rollout fixtures
def gen_rollout(su):
    tensors, meta = {}, {"cases": []}
    B, Hh, Ww, C = 1, 4, 6, 64
    N = Hh * Ww
    g = torch.Generator().manual_seed(314)
    z0 = torch.randn(B, N, C, generator=g).to(torch.bfloat16)
    ehs = (0.1 * torch.randn(B, 8, 32, generator=g)).to(torch.bfloat16)
    pooled = torch.randn(B, 16, generator=g).to(torch.bfloat16)
    text_ids = torch.zeros(B, 3)
    ids = torch.zeros(Hh, Ww, 3)
    ids[..., 1] += torch.arange(Hh)[:, None]
    ids[..., 2] += torch.arange(Ww)[None, :]
    ids = ids.reshape(N, 3)
    tensors.update({"in/z0": z0, "in/ehs": ehs, "in/pooled": pooled, "in/text_ids": text_ids, "in/img_ids": ids})
    model = ToyTransformer(C, seed=3)

    def run(tag, T, window, **kw):
        a = dpm_args(**kw)
        sig = su.sd3_time_shift(a.shift, torch.linspace(1, 0, T + 1))
        det = [True] * T
        for i in window:
            det[i] = False
        NOISE_LOG.clear()
        torch.manual_seed(4242)
        real_randn_like = torch.randn_like

        def logged_randn_like(t, *aa, **kk):           # dance_grpo_step draws with randn_like, not randn_tensor
            r = real_randn_like(t, *aa, **kk)
            NOISE_LOG.append(r.clone())
            return r

        torch.randn_like = logged_randn_like
        try:
            with torch.no_grad():
                z, lat, all_lat, all_lp = su.run_sample_step(a, z0, range(T), sig, model, ehs, pooled, text_ids, ids,
                                                             True, det)
        finally:
            torch.randn_like = real_randn_like
        tensors[f"{tag}/z"] = z.clone()
        tensors[f"{tag}/latents"] = lat.clone()
        tensors[f"{tag}/all_latents"] = all_lat.clone()
        tensors[f"{tag}/all_log_probs"] = all_lp.clone()
        for k, nz in enumerate(NOISE_LOG):
            tensors[f"{tag}/noise{k}"] = nz.clone()
        meta["cases"].append({"tag": tag, "T": T, "window": list(window), "n_noise": len(NOISE_LOG),
                              "args": vars(a), "steps_run": int(all_lp.shape[1])})

    run("mix_T8_w23", 8, [2, 3])
    run("mix_T8_w01", 8, [0, 1])
    run("mix_T8_drop", 8, [1, 2], drop_last_sample=True)
    run("dance_T8_w12", 8, [1, 2], flow_grpo_sampling=False, eta=0.3)
    run("flash_T25_w01", 25, [0, 1], dpm_algorithm_type="dpmsolver++", dpm_apply_strategy="post",
        dpm_post_compress_ratio=0.4, dpm_solver_order=2, dpm_solver_type="midpoint")
    run("flash_T25_w10", 25, [10, 11, 12, 13], dpm_algorithm_type="dpmsolver++", dpm_apply_strategy="post",
        dpm_post_compress_ratio=0.4, dpm_solver_order=2, dpm_solver_type="midpoint")
    run("flash_T25_w21", 25, [21, 22], dpm_algorithm_type="dpmsolver++", dpm_apply_strategy="post",
        dpm_post_compress_ratio=0.2, dpm_solver_order=2, dpm_solver_type="midpoint")
    run("flash3_T12_w23", 12, [2, 3], dpm_algorithm_type="dpmsolver++", dpm_apply_strategy="post",
        dpm_post_compress_ratio=0.6, dpm_solver_order=3, dpm_solver_type="heun")
    run("dpmall_T8", 8, [2, 3], dpm_algorithm_type="dpmsolver++", dpm_apply_strategy="all",
        dpm_solver_order=2, dpm_solver_type="midpoint")
    return tensors, meta


# --------------------------------------------------------------------------- window scheduler
def gen_windows(gs):
    out = []

    def trace(n_iter, seeds=None, **kw):
        st = gs.GRPOTrainingStates(**kw)
        seq = []
        for it in range(n_iter):
            seq.append([int(t) for t in st.get_current_timesteps()])
            st.update_iteration(seed=None if seeds is None else seeds + it)
        out.append({"params": kw, "seeds": seeds, "n_iter": n_iter, "sequence": seq,
                    "final": {"cur_timestep": int(st.cur_timestep), "cur_iter_in_group": int(st.cur_iter_in_group)}})

    trace(40, iters_per_group=2, group_size=2, max_timesteps=6, prog_overlap=True, prog_overlap_step=1, roll_back=True)
    trace(40, iters_per_group=3, group_size=4, max_timesteps=23, prog_overlap=True, prog_overlap_step=1, roll_back=True)
    trace(40, iters_per_group=3, group_size=4, max_timesteps=23, prog_overlap=False, roll_back=False)
    trace(30, iters_per_group=2, group_size=4, max_timesteps=23, prog_overlap=False, roll_back=True)
    trace(30, iters_per_group=2, group_size=4, max_timesteps=23, prog_overlap=True, prog_overlap_step=0, roll_back=True)
    trace(60, iters_per_group=25, group_size=4, max_timesteps=23, prog_overlap=True, prog_overlap_step=1, roll_back=True)
    trace(80, iters_per_group=8, group_size=2, max_timesteps=10, sample_strategy="decay", prog_overlap=True,
          prog_overlap_step=1, roll_back=True)
    trace(80, iters_per_group=8, group_size=3, max_timesteps=14, sample_strategy="decay", max_iters_per_group=10,
          min_iters_per_group=3, roll_back=False)
    trace(120, iters_per_group=5, group_size=4, max_timesteps=23, sample_strategy="exp_decay", prog_overlap=True,
          prog_overlap_step=1, roll_back=True)
    trace(60, iters_per_group=5, group_size=2, max_timesteps=23, sample_strategy="exp_decay",
          exp_decay_thre_timestep=4, exp_decay_k=0.35, roll_back=True)
    trace(40, seeds=100, iters_per_group=1, group_size=4, max_timesteps=23, sample_strategy="random")
    trace(10, iters_per_group=2, group_size=2, max_timesteps=23, cur_timestep=5, prog_overlap=True, roll_back=True)
    return out


# --------------------------------------------------------------------------- trainer fixtures
class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Serves inert packages for third-party top-levels that are absent offline."""
    ROOTS = ("diffusers", "cv2", "wandb", "peft", "HPSv2", "ImageReward", "torchvision", "loguru", "open_clip",
             "flash_attn", "liger_kernel", "decord", "torch_xla", "hpsv2", "clip", "bitsandbytes", "moviepy",
             "imageio", "av", "timm", "ftfy", "deepspeed", "xformers", "triton", "apex", "sklearn_stub")

    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in self.ROOTS and fullname not in sys.modules:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


class _StubMeta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _make_stub(f"{cls.__name__}.{name}")


def _make_stub(name):
    def _call(cls, *a, **k):
        if len(a) == 1 and not k and (isinstance(a[0], type) or callable(a[0])):
            return a[0]  # decorator pass-through
        return object.__new__(cls)
    return _StubMeta(name.split(".")[-1], (), {"__new__": _call, "__init__": lambda self, *a, **k: None,
                                               "__getattr__": lambda self, n: _make_stub(n),
                                               "__call__": lambda self, *a, **k: (a[0] if len(a) == 1 and not k and callable(a[0])
                                                                                 and not isinstance(a[0], type) else self)})


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name == "randn_tensor":
            return _randn_tensor
        v = _make_stub(name)
        setattr(self, name, v)
        return v


def load_trainer():
    import transformers.pipelines  # noqa: F401  real ones first
    import accelerate.utils  # noqa: F401
    for k in [k for k in sys.modules if k.split(".")[0] == "diffusers"]:
        del sys.modules[k]
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, REF)
    import fastvideo.train_grpo_flux as tg
    return tg


class _DummyVae:
    def enable_tiling(self):
        pass

    def decode(self, latents, return_dict=False):
        return (latents,)


class _DummyProc:
    def __init__(self, *a, **k):
        pass

    def postprocess(self, image):
        return [_DummyImage()]


class _DummyImage:
    def save(self, path):
        pass


class _DummySched:
    def step(self):
        pass


def gen_trainer(tg):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29581")
    os.environ["RANK"] = "0"
    os.environ["LOCAL_RANK"] = "0"
    os.environ["WORLD_SIZE"] = "1"
    dist.init_process_group("gloo", rank=0, world_size=1)
    tensors, meta = {}, {"cases": []}
    tmp = "/tmp/mixgrpo_fixture_out"

    def run(tag, rewards_per_head, weights, T=8, window=(2, 3), G=4, opt_lr=1e-2, **kw):
        heads = list(rewards_per_head.keys())
        counter = {"i": 0}

        def fake_compute_reward(images, prompts, reward_function, reward_weights):
            i = counter["i"]
            counter["i"] += 1
            rd = {h: [float(rewards_per_head[h][i])] for h in heads}
            tot = [sum(float(weights[h]) * rd[h][0] for h in heads)]
            return tot, [1], rd, {h: [1] for h in heads}

        tg.compute_reward = fake_compute_reward
        tg.VaeImageProcessor = _DummyProc
        base = dict(w=48, h=32, t=1, sampling_steps=T, shift=3.0, init_same_noise=True, training_strategy="part",
                    output_dir=tmp, experiment_name=tag, reward_model="toy", multi_reward_mix="advantage_aggr",
                    use_group=True, num_generations=G, trimmed_ratio=0.0, advantage_rerange_strategy="null",
                    clip_range=1e-4, adv_clip_max=5.0, kl_coeff=0.0, gradient_accumulation_steps=2,
                    frozen_init_timesteps=-1, timestep_fraction=1.0,
                    dpm_algorithm_type="null", dpm_apply_strategy="post", dpm_post_compress_ratio=0.4,
                    dpm_solver_order=2, dpm_solver_type="midpoint", sample_strategy="progressive",
                    flow_grpo_sampling=True, eta=0.7, drop_last_sample=False)
        base.update(kw)
        a = Namespace(**base)
        os.makedirs(f"{tmp}/{a.training_strategy}_{tag}", exist_ok=True)
        model = ToyTransformer(64, seed=11)
        opt = torch.optim.AdamW(model.parameters(), lr=opt_lr, betas=(0.9, 0.999), weight_decay=1e-4, eps=1e-8)
        g = torch.Generator().manual_seed(77)
        ehs = (0.1 * torch.randn(1, 8, 32, generator=g)).to(torch.bfloat16)
        pooled = torch.randn(1, 16, generator=g).to(torch.bfloat16)
        text_ids = torch.zeros(1, 3)
        loader = iter([(ehs, pooled, text_ids, ["a toy prompt"])])
        NOISE_LOG.clear()
        torch.manual_seed(714)
        import random as _r
        _r.seed(714)
        res = tg.train_one_step(a, torch.device("cpu"), model, _DummyVae(), None, opt, _DummySched(), loader, None,
                                1.0, list(window), 0, weights)
        tensors[f"{tag}/ehs"] = ehs
        tensors[f"{tag}/pooled"] = pooled
        for n, p in model.named_parameters():
            tensors[f"{tag}/param_after/{n}"] = p.detach().clone()
        rd = res[5]
        meta["cases"].append({"tag": tag, "args": {k: v for k, v in vars(a).items()}, "window": list(window),
                              "rewards": {h: [float(x) for x in rewards_per_head[h]] for h in heads},
                              "weights": {h: float(weights[h]) for h in heads}, "opt_lr": opt_lr,
                              "ret": {"total_loss": float(res[0]), "grad_norm": float(res[1]),
                                      "policy_total_loss": float(res[2]), "kl_total_loss": float(res[3]),
                                      "total_clip_frac": float(res[4]),
                                      "reward_mean": rd if not isinstance(rd, dict) else {k: float(v) for k, v in rd.items()}}})

    run("adv_single", {"HeadA": [0.1, 0.2, 0.3, 0.4]}, {"HeadA": 1.0})
    run("adv_multi", {"HeadA": [0.1, 0.5, 0.3, 0.9], "HeadB": [2.0, 1.0, 4.0, 3.0]}, {"HeadA": 1.0, "HeadB": 0.5},
        kl_coeff=0.01)
    run("adv_trim", {"HeadA": [0.9, 0.2, 0.35, 0.4, 0.1, 0.77]}, {"HeadA": 1.0}, G=6, trimmed_ratio=0.25,
        gradient_accumulation_steps=3)
    run("reward_aggr", {"HeadA": [0.3, 0.1, 0.8, 0.4]}, {"HeadA": 1.0}, multi_reward_mix="reward_aggr")
    run("nogroup", {"HeadA": [0.3, 0.1, 0.8, 0.4]}, {"HeadA": 1.0}, multi_reward_mix="reward_aggr", use_group=False,
        G=1, gradient_accumulation_steps=1)
    run("const_reward", {"HeadA": [0.5, 0.5, 0.5, 0.5]}, {"HeadA": 1.0})
    run("balance", {"HeadA": [0.1, 0.9, 0.5, 0.5, 0.2, 0.95]}, {"HeadA": 1.0}, G=6, advantage_rerange_strategy="balance",
        gradient_accumulation_steps=2)
    run("dance_all", {"HeadA": [0.1, 0.2, 0.3, 0.4]}, {"HeadA": 1.0}, training_strategy="all", flow_grpo_sampling=False,
        eta=0.3, timestep_fraction=0.6)
    run("flash_post", {"HeadA": [0.4, 0.2, 0.3, 0.1]}, {"HeadA": 1.0}, T=12, window=(0, 1),
        dpm_algorithm_type="dpmsolver++", dpm_post_compress_ratio=0.4)
    # DPM-Solver++ on EVERY step (dpm_apply_strategy="all"): SDE dpm_step inside the window during the rollout, and a
    # state-less first-order SDE dpm_step in the replay whose log-prob differentiates through its mean (:170-180)
    run("dpm_all", {"HeadA": [0.4, 0.3, 0.1, 0.2]}, {"HeadA": 1.0}, dpm_algorithm_type="dpmsolver++", dpm_apply_strategy="all",
        dpm_solver_order=2, dpm_solver_type="midpoint", kl_coeff=0.01)
    dist.destroy_process_group()
    return tensors, meta


# --------------------------------------------------------------------------- MMDiT structural witnesses
def gen_mmdit_witness():
    """The FLUX MMDiT itself is third-party (diffusers, absent).  What /root/reference DOES hold is sibling code built
    from the same formulas (witness table): RoPE tables and rotation, the sinusoidal embedding, RMSNorm,
    modulate / gate.  Their outputs at FLUX shapes (axes 16/56/56, head_dim 128, 256-wide sinusoid) pin the corresponding
    pieces of oracle/mmdit.py; the block wiring stays unpinned."""
    pe = load_ref("ref_posemb_layers", "fastvideo/models/hunyuan/modules/posemb_layers.py")
    nl = load_ref("ref_norm_layers", "fastvideo/models/hunyuan/modules/norm_layers.py")
    mn = load_ref("ref_mochi_norm", "fastvideo/models/mochi_hf/norm.py")
    ml = load_ref("ref_modulate_layers", "fastvideo/models/hunyuan/modules/modulate_layers.py")
    # embed_layers.py does `from ..utils.helpers import to_2tuple` at import: give it its own package context
    pkg = types.ModuleType("ref_hy")
    pkg.__path__ = [os.path.join(REF, "fastvideo/models/hunyuan")]
    sys.modules["ref_hy"] = pkg
    for sub in ("modules", "utils"):
        m = types.ModuleType(f"ref_hy.{sub}")
        m.__path__ = [os.path.join(REF, "fastvideo/models/hunyuan", sub)]
        sys.modules[f"ref_hy.{sub}"] = m
    load_ref("ref_hy.utils.helpers", "fastvideo/models/hunyuan/utils/helpers.py")
    el = load_ref("ref_hy.modules.embed_layers", "fastvideo/models/hunyuan/modules/embed_layers.py")

    t, meta = {}, {}
    g = torch.Generator().manual_seed(20251004)
    # --- RoPE tables: per axis, positions 0..63 (FLUX ids: text rows all 0, image rows (0, row, col), 64 x 64 grid)
    axes = (16, 56, 56)
    pos = torch.arange(64).float()
    for a, dim in enumerate(axes):
        cos, sin = pe.get_1d_rotary_pos_embed(dim, pos, theta=10000.0, use_real=True)
        t[f"rope/axis{a}/cos"], t[f"rope/axis{a}/sin"] = cos.contiguous(), sin.contiguous()
    meta["rope"] = {"axes_dims": list(axes), "positions": 64, "theta": 10000.0,
                    "note": "witness computes the frequencies in float32 (posemb_layers.py:303-306); diffusers' FluxPosEmbed "
                            "(and oracle/mmdit.py) in float64 on CPU/CUDA"}
    # --- rotation: interleaved pairs, fp32 arithmetic, cast back to the input dtype
    S_, H_, D_ = 64, 2, 128
    ids = torch.stack([torch.zeros(S_), torch.arange(S_) % 64, (torch.arange(S_) * 7) % 64], dim=1)
    cos = torch.cat([t[f"rope/axis{a}/cos"][ids[:, a].long()] for a in range(3)], dim=1)
    sin = torch.cat([t[f"rope/axis{a}/sin"][ids[:, a].long()] for a in range(3)], dim=1)
    t["rot/ids"], t["rot/cos"], t["rot/sin"] = ids, cos, sin
    for tag, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        xq = torch.randn(1, S_, H_, D_, generator=g).to(dt)
        xk = torch.randn(1, S_, H_, D_, generator=g).to(dt)
        oq, ok = pe.apply_rotary_emb(xq, xk, (cos, sin), head_first=False)
        t[f"rot/{tag}/xq"], t[f"rot/{tag}/xk"], t[f"rot/{tag}/oq"], t[f"rot/{tag}/ok"] = xq, xk, oq.contiguous(), ok.contiguous()
    # --- sinusoidal timestep embedding [cos | sin], 128 frequencies
    ts = torch.tensor([1000.0, 986.0, 952.0, 971.0, 111.0, 0.0, 3500.0, 3488.0, 123.5])
    t["sincos/t"], t["sincos/out"] = ts, el.timestep_embedding(ts, 256)
    # --- RMSNorm over head_dim 128 with a learned weight (q / k norms)
    w = 1.0 + 0.1 * torch.randn(128, generator=g)
    for tag, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        x = (2.0 * torch.randn(2, 3, 16, 128, generator=g)).to(dt)
        hy = nl.RMSNorm(128, eps=1e-6)
        mo = mn.MochiRMSNorm(128, eps=1e-6)
        with torch.no_grad():
            hy.weight.copy_(w)
            mo.weight.copy_(w)
            t[f"rms/{tag}/x"], t[f"rms/{tag}/hunyuan"], t[f"rms/{tag}/mochi"] = x, hy(x).contiguous(), mo(x).contiguous()
    t["rms/w"] = w
    meta["rms"] = {"eps": 1e-6, "note": "hunyuan RMSNorm rounds the normalised tensor to the input dtype BEFORE the weight "
                                        "multiply (norm_layers.py:56-59); MochiRMSNorm multiplies in fp32 and casts at the end "
                                        "(mochi_hf/norm.py:52-63), like diffusers' RMSNorm with an fp32 weight"}
    # --- modulate / gate as AdaLN-Zero uses them under autocast: fp32 LayerNorm output, bf16 shift / scale / gate
    x = torch.randn(2, 6, 3072, generator=g)
    ln = torch.nn.functional.layer_norm(x, (3072,), None, None, 1e-6)
    shift = (0.2 * torch.randn(2, 3072, generator=g)).to(torch.bfloat16)
    scale = (0.2 * torch.randn(2, 3072, generator=g)).to(torch.bfloat16)
    gate = (0.5 * torch.randn(2, 3072, generator=g)).to(torch.bfloat16)
    y = torch.randn(2, 6, 3072, generator=g).to(torch.bfloat16)
    res = torch.randn(2, 6, 3072, generator=g).to(torch.bfloat16)
    t["mod/x"], t["mod/shift"], t["mod/scale"] = x, shift, scale
    t["mod/out"] = ml.modulate(ln, shift=shift, scale=scale).contiguous()
    t["gate/y"], t["gate/gate"], t["gate/res"] = y, gate, res
    t["gate/out"] = (res + ml.apply_gate(y, gate=gate)).contiguous()
    meta["dtypes"] = {k: str(v.dtype) for k, v in t.items()}
    return t, meta


# --------------------------------------------------------------------------- inference sampler schedule
def gen_sampler():
    """The two schedule helpers the reference VENDORS as plain Python (fastvideo/models/flux_hf/pipeline_flux.py:73-84,
    87-145; call site fastvideo/sample/sample_flux.py:248-264): `calculate_shift` values, and what `retrieve_timesteps`
    hands to the scheduler (recorded by a fake scheduler: the unshifted sigma grid, mu, and what it returns).  The
    scheduler's own shifting / stepping lives in diffusers (absent) and stays unpinned."""
    # the real transformers classes the module imports, BEFORE the stub finder (it would serve a version-less torchvision)
    from transformers import (CLIPImageProcessor, CLIPTextModel, CLIPTokenizer, CLIPVisionModelWithProjection,  # noqa: F401
                              T5EncoderModel, T5TokenizerFast)
    for k in [k for k in sys.modules if k.split(".")[0] == "diffusers"]:
        del sys.modules[k]
    if not any(isinstance(f, _StubFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _StubFinder())
    pf = load_ref("ref_pipeline_flux", "fastvideo/models/flux_hf/pipeline_flux.py")
    out = {"calculate_shift": {}, "retrieve_timesteps": []}
    for n in (256, 512, 1024, 1536, 2025, 2304, 3072, 4096):
        out["calculate_shift"][str(n)] = {"default": pf.calculate_shift(n),
                                          "flux_config": pf.calculate_shift(n, 256, 4096, 0.5, 1.15)}

    class Recorder:
        order = 1

        def set_timesteps(self, num_inference_steps=None, device=None, sigmas=None, mu=None):
            self.seen = dict(num_inference_steps=num_inference_steps, device=device,
                             sigmas=None if sigmas is None else [float(x) for x in sigmas], mu=mu)
            self.timesteps = torch.arange(len(sigmas) if sigmas is not None else num_inference_steps)

    for T, n_img in ((28, 4096), (50, 1024), (8, 256)):
        sig = np.linspace(1.0, 1 / T, T)                    # sample_flux.py:249
        mu = pf.calculate_shift(n_img, 256, 4096, 0.5, 1.15)
        r = Recorder()
        ts, nst = pf.retrieve_timesteps(r, T, "cpu", sigmas=sig, mu=mu)
        out["retrieve_timesteps"].append({"T": T, "n_img": n_img, "mu": mu, "passed": r.seen, "returned_len": len(ts),
                                          "returned_steps": int(nst)})
    return out


# --------------------------------------------------------------------------- VAE tiling
def vae_tile_standin(tile, up):
    """The recording decoder stand-in of the tiling fixtures (a deterministic function of the latent tile, so that a test
    can hand the SAME function to the oracle / the product): channels 0..2, nearest `up` x in H and W, plus 1/8 of the
    tile-local row index, rounded to bf16 like a bf16 decoder's output."""
    x = tile[:, :3].float()
    x = x.repeat_interleave(up, dim=-2).repeat_interleave(up, dim=-1)
    rows = torch.arange(x.shape[-2], dtype=torch.float32).view(*([1] * (x.dim() - 2)), -1, 1)
    return (x + rows / 8).to(torch.bfloat16)


def gen_vae_tiling():
    """What the reference HOLDS of the VAE decode as plain Python, run in place
    (fastvideo/models/hunyuan/vae/autoencoder_kl_causal_3d.py): the tile-size rule of `__init__` (:132-139, the reference's
    own constructor on small channel counts), `blend_v` / `blend_h` (:384-399) on 5-D [B, C, T, H, W] tensors, and
    `spatial_tiled_decode` (:472-525) with an identity `post_quant_conv` and the recording stand-in above as `decoder`.
    The convolutions, GroupNorm and attention of the decoder are diffusers code (absent) and stay unpinned."""
    for k in [k for k in sys.modules if k.split(".")[0] == "diffusers"]:
        del sys.modules[k]
    if not any(isinstance(f, _StubFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _StubFinder())
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import fastvideo.models.hunyuan.vae.autoencoder_kl_causal_3d as A
    C = A.AutoencoderKLCausal3D

    class Holder(C):
        pass

    t, meta = {}, {"tile_sizes": [], "blend": [], "tiled": []}
    for sample_size, nblk in ((1024, 4), (512, 4), (256, 3), (32, 2), (96, 3)):
        h = Holder.__new__(Holder)
        h.config = Namespace(sample_size=sample_size, block_out_channels=(8,) * nblk)
        C.__init__(h, in_channels=3, out_channels=3, block_out_channels=(8,) * nblk, layers_per_block=1, latent_channels=4,
                   norm_num_groups=4, sample_size=sample_size, sample_tsize=64, mid_block_add_attention=False)
        meta["tile_sizes"].append({"sample_size": sample_size, "n_blocks": nblk, "tile_sample_min_size": h.tile_sample_min_size,
                                   "tile_latent_min_size": h.tile_latent_min_size, "tile_overlap_factor": h.tile_overlap_factor})
    g = torch.Generator().manual_seed(20251005)
    h = Holder.__new__(Holder)
    for i, (dt, shape_a, shape_b, extent) in enumerate(((torch.bfloat16, (2, 3, 1, 12, 10), (2, 3, 1, 12, 10), 8),
                                                        (torch.bfloat16, (1, 3, 1, 6, 16), (1, 3, 1, 9, 16), 100),
                                                        (torch.float32, (1, 2, 1, 16, 16), (1, 2, 1, 16, 16), 4),
                                                        (torch.bfloat16, (1, 3, 1, 256, 8), (1, 3, 1, 256, 8), 256))):
        for kind in ("v", "h"):
            sa = shape_a if kind == "v" else shape_a[:3] + (shape_a[4], shape_a[3])
            sb = shape_b if kind == "v" else shape_b[:3] + (shape_b[4], shape_b[3])
            a = torch.randn(sa, generator=g).to(dt)
            b = torch.randn(sb, generator=g).to(dt)
            b_in = b.clone()
            out = (C.blend_v if kind == "v" else C.blend_h)(h, a, b, extent)
            assert out is b                                           # in place on the later tile
            key = f"blend/{i}/{kind}"
            t[key + "/a"], t[key + "/b"], t[key + "/out"] = a, b_in, out.clone()
            meta["blend"].append({"key": key, "extent": extent, "dtype": str(dt)})
    for i, (tl, ts, zshape) in enumerate(((16, 32, (1, 4, 1, 20, 28)), (16, 32, (2, 4, 1, 16, 40)), (8, 64, (1, 4, 1, 13, 9)),
                                          (16, 128, (1, 4, 1, 33, 16)))):
        h = Holder.__new__(Holder)
        h.tile_latent_min_size, h.tile_sample_min_size, h.tile_overlap_factor = tl, ts, 0.25
        calls = []
        h.post_quant_conv = lambda x: x

        def dec(tile, _up=ts // tl, _calls=calls):
            _calls.append([int(tile.shape[-2]), int(tile.shape[-1])])
            return vae_tile_standin(tile, _up)
        h.decoder = dec
        z = torch.randn(zshape, generator=g)
        out = C.spatial_tiled_decode(h, z, return_dict=False)[0]
        key = f"tiled/{i}"
        t[key + "/z"], t[key + "/out"] = z, out.contiguous()
        meta["tiled"].append({"key": key, "tile_latent_min_size": tl, "tile_sample_min_size": ts, "tile_overlap_factor": 0.25,
                              "decoder_calls": calls, "out_shape": list(out.shape)})
    return t, meta


def main():
    which = sys.argv[1:] or ["solver", "rollout", "windows", "trainer", "mmdit", "sampler", "vae"]
    install_light_stub()
    su = load_ref("ref_sampling_utils", "fastvideo/utils/sampling_utils.py")
    if "solver" in which:
        t, m = gen_solver(su)
        save_file({k: v.contiguous() for k, v in t.items()}, os.path.join(HERE, "solver_steps.safetensors"))
        json.dump(m, open(os.path.join(HERE, "solver_steps.json"), "w"), indent=1)
        print("solver:", len(t), "tensors")
    if "rollout" in which:
        t, m = gen_rollout(su)
        save_file({k: v.contiguous() for k, v in t.items()}, os.path.join(HERE, "rollout.safetensors"))
        json.dump(m, open(os.path.join(HERE, "rollout.json"), "w"), indent=1)
        print("rollout:", len(t), "tensors")
    if "windows" in which:
        gs = load_ref("ref_grpo_states", "fastvideo/utils/grpo_states.py")
        json.dump(gen_windows(gs), open(os.path.join(HERE, "windows.json"), "w"))
        print("windows ok")
    if "mmdit" in which:
        t, m = gen_mmdit_witness()
        save_file({k: v.contiguous() for k, v in t.items()}, os.path.join(HERE, "mmdit_witness.safetensors"))
        json.dump(m, open(os.path.join(HERE, "mmdit_witness.json"), "w"), indent=1)
        print("mmdit witness:", len(t), "tensors")
    if "sampler" in which:
        json.dump(gen_sampler(), open(os.path.join(HERE, "sampler_schedule.json"), "w"), indent=1)
        print("sampler schedule ok")
    if "vae" in which:
        t, m = gen_vae_tiling()
        save_file({k: v.contiguous() for k, v in t.items()}, os.path.join(HERE, "vae_tiling.safetensors"))
        json.dump(m, open(os.path.join(HERE, "vae_tiling.json"), "w"), indent=1)
        print("vae tiling:", len(t), "tensors")
    if "trainer" in which:
        tg = load_trainer()
        t, m = gen_trainer(tg)
        save_file({k: v.contiguous() for k, v in t.items()}, os.path.join(HERE, "trainer.safetensors"))
        json.dump(m, open(os.path.join(HERE, "trainer.json"), "w"), indent=1, default=str)
        print("trainer:", len(t), "tensors")


if __name__ == "__main__":
    main()
