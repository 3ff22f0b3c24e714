#!/usr/bin/env python3
"""GRPO train-step throughput (images/sec) of the MI355X-native MixGRPO engine on FLUX.1-dev 1024^2.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" is one `train_one_step` (BASELINE.json metric): group rollout (G x T MMDiT forwards + solver steps),
reward gather, group-relative advantages, G x W replayed forward/backward passes, optimizer steps every `accum`
samples.  Synthetic inputs: random-init FLUX.1-dev weights (N(0, 0.02^2)), 1024x1024 latents,
cached-text-embedding-shaped random tensors, uniform synthetic rewards.  Excluded: VAE decode,
reward-model inference, image/wandb logging, checkpointing.  One rank per GPU, weak scaling (one prompt group per
rank per step, like the reference's DistributedSampler partitioning).

`python bench.py --gpus N` with N > 1 and no launcher (WORLD_SIZE unset) starts the N ranks itself: the parent, BEFORE
any GPU call, runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same flags>` as a child
process, relays its output (rank 0's JSON line) and exits with its code.  Nothing is ever re-exec'ed.

Prints ONE JSON line on rank 0, with `roofline` (bf16 MFMA GEMM family, measured with HIP events around every GEMM
launch of the LAST warm-up step -- an extra untimed step only with --warmup 0) and `cpu_baseline` (the CPU oracle timed
on this box's host cores on a bounded sample).  A progress line per step goes to stderr.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA (guides/MI355X_MICROARCH.md)
F_FWD_1024 = 74.38e12              # algorithmic FLOPs of one FLUX.1-dev forward at 1024^2

WORKLOADS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on
    "flux1dev_1024_T25_W4_G8": dict(h=1024, w=1024, sampling_steps=25, window=4, num_generations=8,
                                    gradient_accumulation_steps=3, layers=(19, 38), heads=24, txt=512),
    # BASELINE.json configs[2]: group of 12, three reward heads (HPSv2 + ImageReward + PickScore stand-ins, weights 1.0,
    # advantage_aggr); reference script finetune_flux_grpo_MixGRPO.sh:150,62
    "flux1dev_1024_T25_W4_G12_3heads": dict(h=1024, w=1024, sampling_steps=25, window=4, num_generations=12,
                                            gradient_accumulation_steps=3, layers=(19, 38), heads=24, txt=512,
                                            reward_heads=("HPSClipRewardModel", "ImageRewardModel", "PickScoreRewardModel")),
    # BASELINE.json configs[3], MixGRPO-Flash: DPM-Solver++ order 2 midpoint outside a 2-step SDE window, post-window
    # schedule compressed by 0.4 (finetune_flux_grpo_MixGRPO_Flash.sh:57,75); window [0, 1] -> 10 solver steps
    "flux1dev_1024_flash_dpmpp2_W2_r0.4": dict(h=1024, w=1024, sampling_steps=25, window=2, num_generations=8,
                                               gradient_accumulation_steps=3, layers=(19, 38), heads=24, txt=512,
                                               args=dict(dpm_algorithm_type="dpmsolver++", dpm_apply_strategy="post",
                                                         dpm_post_compress_ratio=0.4, dpm_solver_order=2,
                                                         dpm_solver_type="midpoint")),
    # BASELINE.json configs[4]: 50-step sampler, sliding 4-step window, group of 16, e4m3 MFMA attention forward
    "flux1dev_1024_T50_W4_G16_fp8attn": dict(h=1024, w=1024, sampling_steps=50, window=4, num_generations=16,
                                             gradient_accumulation_steps=3, layers=(19, 38), heads=24, txt=512,
                                             attention="fp8"),
    # a small stand-in for quick checks (NOT the metric): 2+2 blocks, 256^2
    "tiny_256_T8_W2_G4": dict(h=256, w=256, sampling_steps=8, window=2, num_generations=4,
                              gradient_accumulation_steps=2, layers=(2, 2), heads=24, txt=64),
}


def flops_per_forward(cfg, n_img, n_txt):
    d = cfg.dim
    S = n_img + n_txt
    lin = 0
    lin += cfg.num_layers * (12 * d * d * S + 2 * 6 * d * d)          # qkv+out (4 d^2) + ff (8 d^2) per token, both streams
    lin += cfg.num_single_layers * (12 * d * d * S + 3 * d * d)      # qkv (3) + mlp (4) + out (5) = 12 d^2 per token
    attn = (cfg.num_layers + cfg.num_single_layers) * 2 * S * S * d
    emb = n_img * cfg.in_channels * d * 2 + n_txt * cfg.joint_attention_dim * d
    return 2.0 * (lin + attn + emb)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: two untimed steps first -- the SECOND train step of a process allocates the keep buffers' growth (114 device
    # allocations, 0.3-1.2 s on a cold box: `allocator_in_timed_region`), so one warm-up step would put it in the timed region
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="flux1dev_1024_T25_W4_G8", choices=list(WORKLOADS))
    ap.add_argument("--train-microbatch", type=int, default=4,
                    help="replayed (sample, step) pairs per forward/backward.  4 -> micro-batches of 4+4+4 (and 4+4 for the "
                         "leftover chunk): at 4 the FF pre-activations AND the QKV outputs of all 57 blocks fit in HBM beside the "
                         "optimizer state, so the recompute pass runs no GEMM at all (flux_backward.KEEP_FF / KEEP_QKV); same "
                         "box, round 4: 18.45 s per step against 18.72 s at 7 (7+5 / 7+1, 18 blocks kept)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vae", action="store_true", help="skip the VAE-decode side measurement (reported beside the metric)")
    ap.add_argument("--skip-dead-backward", action="store_true",
                    help="do not execute the backward passes of the G %% accum leftover samples, whose gradients the "
                         "reference computes and discards (identical outputs; NOT the default, NOT the headline number)")
    ap.add_argument("--attention", default="bf16", choices=["bf16", "fp8"],
                    help="attention forward dtype; fp8 = the e4m3 MFMA path of BASELINE.json configs[4] (NOT the headline "
                         "configuration: the line's dtype then reads bf16+fp8attn)")
    ap.add_argument("--gemm-shapes", default=None, help="write the per-shape GEMM time table of the roofline step here")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0,
                    help="wall-time budget of the CPU-baseline sample (it stops widening the sample once this is used up)")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()

    if a.cpu_baseline_child:        # (internal) the CPU-baseline sample as its own process: no GPU call, one JSON line
        print("CPU_BASELINE " + json.dumps(_cpu_baseline_measure(a.cpu_baseline_seconds)), flush=True)
        return

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the driver's launcher-less form `python3 bench.py --gpus N ...`: start the N ranks as CHILD processes (one per GPU,
        # RCCL) before this process has made any GPU call, relay their output and exit with the launcher's code
        sys.exit(_self_launch(a.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    backend = os.environ.get("MGX_DIST_BACKEND", "nccl") if world > 1 else None    # nccl == RCCL on ROCm
    if world > 1 and backend == "nccl" and a.gpus > torch.cuda.device_count():
        # fail fast, before any collective: RCCL needs one device per rank (two ranks on one device hang in the communicator
        # set-up); `MGX_DIST_BACKEND=gloo` is the rehearsal form on a box with fewer GPUs
        raise SystemExit(f"bench.py: --gpus {a.gpus} over RCCL needs {a.gpus} visible devices, this host has "
                         f"{torch.cuda.device_count()} (rehearse with MGX_DIST_BACKEND=gloo)")
    local_dev = local_rank % torch.cuda.device_count()      # (== local_rank on a full node; lets a 1-GPU box rehearse N ranks over gloo)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        # build the RCCL communicator (rings over xGMI) now: with --warmup 0 it would otherwise be created inside the
        # timed step by the first reward all-gather
        _w = torch.ones(1 << 20, device=dev)
        dist.all_reduce(_w)
        dist.all_gather([torch.empty(4, device=dev) for _ in range(world)], torch.ones(4, device=dev))
        torch.cuda.synchronize()
        del _w
    if a.gpus != world:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")

    from mixgrpo_amd import ops
    from mixgrpo_amd import train_grpo_flux as TG
    from mixgrpo_amd.flux import FluxConfig, FluxTransformer2DModel
    from mixgrpo_amd.grpo_states import GRPOTrainingStates
    from mixgrpo_amd.optim import ConstantWithWarmup, FusedAdamW

    wl = WORKLOADS[a.workload]
    cfg = FluxConfig(num_layers=wl["layers"][0], num_single_layers=wl["layers"][1], num_attention_heads=wl["heads"])
    if wl.get("attention"):
        a.attention = wl["attention"]
    model = FluxTransformer2DModel(cfg, device=dev, attention_dtype=a.attention).init_synthetic(seed=0, std=0.02)
    opt = FusedAdamW(model, lr=1e-5, betas=(0.9, 0.999), weight_decay=1e-4, eps=1e-8)
    sched = ConstantWithWarmup(opt, 0)
    args = TG.default_args(h=wl["h"], w=wl["w"], sampling_steps=wl["sampling_steps"], num_generations=wl["num_generations"],
                           gradient_accumulation_steps=wl["gradient_accumulation_steps"], train_microbatch=a.train_microbatch,
                           skip_dead_backward=a.skip_dead_backward, **wl.get("args", {}))
    T, G, W = args.sampling_steps, args.num_generations, wl["window"]
    heads = wl.get("reward_heads", ("SyntheticReward",))
    reward_weights = {h: 1.0 for h in heads}
    states = GRPOTrainingStates(iters_per_group=25, group_size=W, max_timesteps=T - 2, prog_overlap=True,
                                prog_overlap_step=1, roll_back=True)
    torch.manual_seed(714 + rank)
    n_img = (args.h // 16) * (args.w // 16)
    L = wl["txt"]

    def loader():
        g = torch.Generator(device=dev).manual_seed(714 + rank)
        while True:
            yield ((0.1 * torch.randn(1, L, cfg.joint_attention_dim, device=dev, generator=g)).bfloat16(),
                   torch.randn(1, cfg.pooled_projection_dim, device=dev, generator=g).bfloat16(),
                   torch.zeros(1, 3, device=dev), ["synthetic prompt"])

    step_no = [0]

    def reward_fn(latents, captions):
        g = torch.Generator().manual_seed(1234 + step_no[0] * 131 + rank)
        per_head = {h: torch.rand(latents.shape[0], generator=g) for h in heads}
        return sum(per_head[h] * reward_weights[h] for h in heads), per_head

    it = loader()

    def one_step():
        window = states.get_current_timesteps()
        states.update_iteration()
        trace = {} if os.environ.get("MGX_BENCH_TRACE") else None      # debugging aid: per-pair log-prob drift to stderr
        out = TG.train_one_step(args, dev, model, None, reward_fn, opt, sched, it, None, 1.0, window, step_no[0],
                                reward_weights, trace=trace)
        if trace is not None and rank == 0:
            lp = trace["log_probs"]
            print(f"[trace] step {step_no[0]} adv {[round(x, 3) for x in trace['advantages'].tolist()]} grad_norms "
                  f"{[round(x.item(), 5) for x in trace.get('grad_norms', [])]}", file=sys.stderr)
            for (pairs, nl), gl in zip(trace["new_log_probs"], trace["g_logp"]):
                old = torch.stack([lp[i, t] for i, t in pairs])
                print(f"[trace]   pairs {pairs}\n[trace]   new-old {[f'{x:.2e}' for x in (nl - old).tolist()]}\n"
                      f"[trace]   g_logp {[f'{x:.2e}' for x in gl.tolist()]}", file=sys.stderr)
        step_no[0] += 1
        t_roll = T
        if "dpmsolver" in args.dpm_algorithm_type and args.dpm_apply_strategy == "post":    # Flash: rebuilt schedule
            from mixgrpo_amd import sampling_utils as SU
            sig = SU.sd3_time_shift(args.shift, torch.linspace(1, 0, T + 1))
            det = [j not in set(window) for j in range(T)]
            t_roll = SU.flash_schedule(sig, det, args.dpm_post_compress_ratio, args.shift)[0].numel() - 1
        return out, window, t_roll

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    t_start = time.perf_counter()

    def progress(kind, i, n, t_step):
        if rank == 0:
            print(f"[bench] {kind} step {i + 1}/{n}: {t_step:.1f} s (elapsed {time.perf_counter() - t_start:.0f} s)",
                  file=sys.stderr, flush=True)

    # Roofline step: HIP events around every GEMM launch of ONE untimed train step -- the last warm-up step, so that the
    # default run pays no extra step (an extra one only with --warmup 0).  EVERY rank runs the same steps (a train step is
    # full of collectives); only rank 0 records events.
    # CPU baseline (rank 0, N = 1): measured by a CHILD process on the host cores while the (untimed) warm-up steps run on the
    # GPU, and collected before the timed region starts -- the default run's wall time stays clear of the driver's limit.
    # (The main process's launch thread shares the cores meanwhile: stated in the sample.)  Without warm-up steps it runs
    # after the timed region, as before.
    cpu_child = None
    if not a.no_cpu_baseline and rank == 0 and world == 1 and a.warmup >= 1:
        import subprocess
        cpu_child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child",
                                      "--cpu-baseline-seconds", str(a.cpu_baseline_seconds)], cwd=ROOT,
                                     stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    cpu_measured = None
    profile_step = None if a.no_roofline else max(0, a.warmup - 1)
    n_untimed = max(a.warmup, 0 if a.no_roofline else 1)
    last = None
    for i in range(n_untimed):
        t1 = time.perf_counter()
        if i == profile_step and rank == 0:
            ops.GEMM_PROFILE = []
        res = one_step()
        torch.cuda.synchronize()
        if i == profile_step and rank == 0:
            prof, ops.GEMM_PROFILE = ops.GEMM_PROFILE, None
        if i < a.warmup:
            last = res
        progress("warm-up" if i < a.warmup else "roofline", i, n_untimed, time.perf_counter() - t1)
    if cpu_child is not None:       # must be over before the timed region (normally long done: ~10 s against W x ~20 s)
        try:
            out_, _ = cpu_child.communicate(timeout=300)
            row = [l for l in out_.splitlines() if l.startswith("CPU_BASELINE ")]
            cpu_measured = json.loads(row[0][len("CPU_BASELINE "):]) if row else None
        except Exception:  # noqa: BLE001 - a failed sample falls back to the in-process measurement after the timed region
            cpu_child.kill()
            cpu_measured = None
    fence()
    ms0 = torch.cuda.memory_stats()
    step_s = []
    t0 = time.perf_counter()
    for i in range(a.steps):
        t1 = time.perf_counter()
        last = one_step()
        step_s.append(time.perf_counter() - t1)
        if rank == 0 and (i + 1) % 2 == 0:  # a progress line every other step, from host time only (no device sync)
            progress("timed", i, a.steps, (time.perf_counter() - t1))
    fence()
    dt = time.perf_counter() - t0
    ms1 = torch.cuda.memory_stats()
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()
    images = world * G * a.steps
    value = images / dt
    f_fwd = flops_per_forward(cfg, n_img, L)
    # algorithmic FLOPs per image (recompute not counted), SURVEY.md 8d: (T' + 3 W) forwards, T' = solver steps of the
    # rollout (T, or the rebuilt Flash schedule's length); with --skip-dead-backward the G % accum leftover images run
    # no backward, and are counted accordingly.  The group's shared first rollout step runs once, at batch 1 (bit-identical
    # rows, tests/test_hip_trainer.py): the T' here counts it G times, the line says so.
    dead = (G % args.gradient_accumulation_steps) if a.skip_dead_backward else 0
    T_roll = last[2] if last else T
    flop_img = (T_roll + 3 * W - 2.0 * W * dead / G) * f_fwd
    shared_first = bool(args.init_same_noise and args.use_group)
    flop_img_executed = flop_img - (f_fwd * (G - 1) / G if shared_first else 0.0)

    roofline = None
    if not a.no_roofline and rank == 0:
        tot_ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in prof)
        tot_fl = sum(f for _, _, f, _ in prof)
        n_launch = len(prof)
        if a.gemm_shapes:
            agg = {}
            for e0, e1, f, shp in prof:
                c = agg.setdefault(shp, [0, 0.0, 0.0])
                c[0] += 1
                c[1] += e0.elapsed_time(e1)
                c[2] += f
            rows = [{"M": k[0], "N": k[1], "K": k[2], "epilogue": k[3], "calls": v[0], "total_ms": round(v[1], 2),
                     "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 1)} for k, v in agg.items()]
            rows.sort(key=lambda r: -r["total_ms"])
            with open(a.gemm_shapes, "w") as fh:
                for r in rows:
                    fh.write(json.dumps(r) + "\n")
        ach = tot_fl / (tot_ms * 1e-3) / 1e12
        # HBM traffic of the dominant kernel is a rocprofv3 PMC measurement (separate --pmc passes, FETCH_SIZE doubled as
        # MI355X_MICROARCH.md prescribes for gfx950): it cannot be taken inside this process, so the committed summary
        # of the round's PMC run is quoted, per launch of the profiled shape (see profiles/README.md)
        traffic, traffic_src = None, None
        pmc_name = next((n for n in ("r04_gemm_pmc.json", "r03_gemm_pmc.json", "r02_gemm_pmc.json", "r01_gemm_pmc_v9.json")
                         if os.path.exists(os.path.join(ROOT, "profiles", n))), None)
        if pmc_name:
            with open(os.path.join(ROOT, "profiles", pmc_name)) as fh:
                pmc = json.load(fh)
            traffic = pmc["traffic_bytes_per_launch"]
            traffic_src = {"file": f"profiles/{pmc_name}", "shape": pmc["shape"],
                           "algorithmic_bytes_per_launch": pmc["algorithmic_bytes_per_launch"],
                           "effective_clock_ghz": pmc["effective_clock_ghz"], "mfma_busy_frac": pmc["mfma_busy_frac"]}
        roofline = {"bound": "mfma", "kernel": "gemm_pp_kernel<EPI> (bf16 MFMA 256x256x64 persistent ping-pong, all epilogues) "
                                                "+ gemm_kernel<EPI> (128x128x64, small shapes)",
                    "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "launches": n_launch,
                    "avg_launch_ms": round(tot_ms / max(1, n_launch), 4),
                    "gemm_share_of_step": round(tot_ms * 1e-3 / (dt / a.steps), 3),
                    "train_step_frac_of_peak": round(flop_img * value / world / (PEAK_BF16_TFLOPS * 1e12), 4)}
    if world > 1:
        dist.barrier()

    cpu = None
    if not a.no_cpu_baseline and rank == 0 and world == 1:       # a reported baseline, N = 1 only
        cpu = cpu_baseline(flop_img, T_roll, a.cpu_baseline_seconds, measured=cpu_measured,
                           how="in a child process beside the untimed warm-up steps" if cpu_measured else "after the timed region")

    if rank == 0:
        line = {"metric": "GRPO train-step images/sec, FLUX.1-dev 1024^2", "value": round(value, 5), "unit": "images/s",
                "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 1),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if a.attention == "bf16" else "bf16+fp8attn",
                "data": "synthetic",
                "config": {"workload": a.workload, "model": "FLUX.1-dev (random-init, 11.9B)" if wl["layers"] == (19, 38)
                           else f"FLUX-like {wl['layers']} blocks", "resolution": f"{args.h}x{args.w}",
                           "sampling_steps": T, "rollout_solver_steps": T_roll, "sde_window": last[1] if last else None,
                           "group_size": G, "reward_heads": list(heads),
                           "solver": ("flow_grpo_sde+euler_ode" if args.dpm_algorithm_type == "null" else
                                      f"flow_grpo_sde window + {args.dpm_algorithm_type} order {args.dpm_solver_order} "
                                      f"{args.dpm_solver_type}, post ratio {args.dpm_post_compress_ratio}"),
                           "grad_accum": args.gradient_accumulation_steps, "global_batch": world * G,
                           "seq_len": n_img + L, "parallelism": f"dp{world}", "train_microbatch": a.train_microbatch,
                           "train_ff_blocks_kept": (model.ff_blocks_kept() if hasattr(model, "ff_blocks_kept") else 0),
                           "train_qkv_blocks_kept": (model.qkv_blocks_kept() if hasattr(model, "qkv_blocks_kept") else 0),
                           # the no-grad forward's attention plumbing (mixgrpo_amd/ops.py switches; all on by default)
                           "rollout_attention": {"q_prescaled": bool(ops.Q_PRESCALE), "vt_from_projection": bool(ops.LINEAR_VT),
                                                 "qk_norm_in_epilogue": bool(ops.LINEAR_QKNORM), "rope_pair_table": bool(ops.ROPE_PAIR_TABLE)},
                           "skip_dead_backward": bool(a.skip_dead_backward),
                           "algorithmic_pflop_per_image": round(flop_img / 1e15, 3),
                           "executed_pflop_per_image": round(flop_img_executed / 1e15, 3)},
                "images_per_sec_per_gpu": round(value / world, 5),
                "mfma_frac_train_step": round(flop_img * value / world / (PEAK_BF16_TFLOPS * 1e12), 4),
                "mfma_frac_train_step_executed": round(flop_img_executed * value / world / (PEAK_BF16_TFLOPS * 1e12), 4),
                "mfma_frac_note": "mfma_frac_train_step prices the reference's (T' + 3W) forwards per image; the engine runs the "
                                  "group's shared first rollout step once at batch 1 (bit-identical rows), i.e. (G-1)/G of one "
                                  "forward per image less: mfma_frac_train_step_executed counts only what ran (recompute never counted)",
                "lib_version": _lib_version(),
                "hbm_peak_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                "hbm_reserved_gib": round(torch.cuda.max_memory_reserved() / 2 ** 30, 1),
                "hbm_free_gib": round(torch.cuda.mem_get_info()[0] / 2 ** 30, 1),
                # allocator traffic INSIDE the timed region (device mallocs / frees / out-of-memory retries of torch's caching
                # allocator) and the host-side time of each timed step (enqueue time: no device sync between steps)
                "allocator_in_timed_region": {k: ms1.get(k2, 0) - ms0.get(k2, 0) for k, k2 in
                                              (("device_alloc", "num_device_alloc"), ("device_free", "num_device_free"),
                                               ("alloc_retries", "num_alloc_retries"), ("ooms", "num_ooms"))},
                "host_s_per_timed_step": [round(x, 2) for x in step_s],
                "last_step": {"loss": last[0][0], "grad_norm": last[0][1], "clip_frac": last[0][4],
                              "note": "grad_norm = norm of the step's last optimizer update, 0.0 when every pair of that "
                                      "chunk is PPO-clipped at clip_range 1e-4 (DESIGN.md section 6)"} if last else None,
                "roofline": roofline, "cpu_baseline": cpu}
        # what the collective library saw (so that an N > 1 record answers "did RCCL see N ranks?" by itself) and every
        # MGX_* environment switch that selects a kernel or a policy in the shipped library: a stale variable is visible here
        red = getattr(model, "_mgx_grad_reducer", None)
        line["dist"] = {"backend": (dist.get_backend() if world > 1 else None),
                        "world_size": (dist.get_world_size() if world > 1 else 1),
                        "rccl_version": (".".join(str(x) for x in torch.cuda.nccl.version()) if world > 1 and backend == "nccl" else None),
                        "grad_dtype": (getattr(red, "mode", None) if red is not None else os.environ.get("MGX_DP_GRAD_DTYPE", "fp32")) if world > 1 else None,
                        "overlap": (bool(getattr(red, "overlap", False)) if red is not None else os.environ.get("MGX_DP_OVERLAP", "0") == "1") if world > 1 else None,
                        "devices_visible": torch.cuda.device_count(),
                        "note": "measured on this run's ranks only; no 8-GPU number is extrapolated anywhere"}
        line["config"]["env_switches"] = {k: v for k, v in sorted(os.environ.items()) if k.startswith("MGX_")}
        line["vae_decode"] = _vae_decode_side_measurement(dev, args, G) if not a.no_vae else None
        line["config"]["reward_seed"] = "1234 + 131 * step + rank (SURVEY 8d names 1234 + step; same distribution)"
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def _vae_decode_side_measurement(dev, args, G):
    """Reported BESIDE the metric, never inside it: the step's reward stage is a synthetic function of the latents (the reward
    networks' weights are not available offline), so the decode the reference runs before its reward models
    (train_grpo_flux.py:279-289) is timed here on its own -- FLUX VAE configuration, random-init weights, one image of the
    workload's resolution, after the timed region."""
    try:
        free_gib = torch.cuda.mem_get_info()[0] / 2 ** 30
        if free_gib < 8.0:           # scores + probabilities of the mid-block attention + activations: ~4 GiB at 1024^2
            return {"skipped": f"only {free_gib:.1f} GiB of device memory free after the timed region"}
        from mixgrpo_amd.vae import AutoencoderKL
        vae = AutoencoderKL(device=dev).init_synthetic(seed=7)
        vae.enable_tiling()
        z = torch.randn(1, 16, args.h // 8, args.w // 8, device=dev)
        vae.decode(z, return_dict=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            vae.decode(z, return_dict=False)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        return {"ms_per_image": round(ms, 2), "ms_per_step_for_the_group": round(ms * G, 1), "resolution": f"{args.h}x{args.w}",
                "note": "not part of `value`: decode + reward models sit behind reward_function(latents, captions)"}
    except Exception as e:  # noqa: BLE001 - a side measurement must not cost the run its bench line
        return {"error": f"{type(e).__name__}: {e}"[:200]}


def _lib_version():
    from mixgrpo_amd import _lib
    return int(_lib.lib().mgx_version())


def _self_launch(n):
    """`python bench.py --gpus N` without a launcher: run the driver's own launch form as a child process.  Called before
    torch.cuda is touched; the parent never initialises the GPU and never exec()s."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC (RCCL across processes on this driver)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env, cwd=ROOT).returncode


def _host_cores():
    """Threads this process may really use: min(affinity, cgroup cpu quota, cpu_count) -- a GPU box reports all of
    the host's logical cores in os.cpu_count() but grants a share of them."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def _cpu_baseline_measure(budget_s):
    """The CPU oracle (a port: kind "port") timed on this box's host cores on a BOUNDED sample of the workload: full-width
    (d = 3072, 24 heads), full-sequence (4096 image + 512 text tokens) MMDiT forwards with 1 double + 1 single block
    (~2.6 TFLOP each), repeated while the wall-time budget lasts (at least one), and a full-size solver step."""
    import torch
    from oracle import mmdit as OM
    from oracle import solver as OS
    t_begin = time.perf_counter()
    cores = _host_cores()
    torch.set_num_threads(cores)
    cfg = OM.FluxConfig(num_layers=1, num_single_layers=1)
    P = OM.init_params(cfg, seed=0)
    N, L = 4096, 512
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, N, 64, generator=g)
    ehs = 0.1 * torch.randn(1, L, 4096, generator=g)
    pooled = torch.randn(1, 768, generator=g)
    ids = torch.zeros(64, 64, 3)
    ids[..., 1] += torch.arange(64)[:, None]
    ids[..., 2] += torch.arange(64)[None]
    ids = ids.reshape(N, 3)
    times = []
    with torch.no_grad():
        while not times or (time.perf_counter() - t_begin + 1.3 * max(times) < budget_s and len(times) < 4):
            t0 = time.perf_counter()
            OM.forward(P, cfg, x, ehs, torch.tensor([0.954]), torch.tensor([3.5]), torch.zeros(L, 3), pooled, ids)
            times.append(time.perf_counter() - t0)
    xs = torch.randn(1, 4096, 64, generator=g)
    v = torch.randn(1, 4096, 64, generator=g).bfloat16()
    sig = OS.sd3_time_shift(3.0, torch.linspace(1, 0, 26))
    OS.flow_grpo_step(v, xs, 0.7, sig, 3, None)
    t0 = time.perf_counter()
    for _ in range(5):
        OS.flow_grpo_step(v, xs, 0.7, sig, 3, None)
    t_solver = (time.perf_counter() - t0) / 5
    return {"t_fwd": min(times), "n_fwd": len(times), "flops_fwd": flops_per_forward(cfg, N, L), "t_solver": t_solver,
            "cores": cores, "tokens": [N, L], "wall_s": time.perf_counter() - t_begin}


def cpu_baseline(flop_img, t_roll, budget_s, measured=None, how="after the timed region"):
    """`cpu_baseline` object of the line: images/s EXTRAPOLATED by algorithmic FLOPs from the bounded sample to the train
    step (a reported baseline, never the target)."""
    m = measured or _cpu_baseline_measure(budget_s)
    rate = m["flops_fwd"] / m["t_fwd"]
    sec_per_image = flop_img / rate + t_roll * m["t_solver"]
    return {"value": round(1.0 / sec_per_image, 8), "unit": "images/s", "cores": m["cores"], "kind": "port",
            "sample": f"oracle MMDiT forward d=3072, 1 double+1 single block, {m['tokens'][0]}+{m['tokens'][1]} tokens, best of "
                      f"{m['n_fwd']}: {m['t_fwd']:.2f}s = {rate / 1e12:.3f} TFLOP/s; full-size solver step "
                      f"{m['t_solver'] * 1e3:.2f} ms; images/s extrapolated by algorithmic FLOPs to the train step "
                      f"({flop_img / 1e15:.3f} PFLOP/image); {m['wall_s']:.1f} s of wall time, {how}"}


if __name__ == "__main__":
    main()
