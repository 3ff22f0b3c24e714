"""The drop-in entry point end to end on the GPU: `python -m mixgrpo_amd.train_grpo_flux` launched with the flag list of
the reference's launcher (scripts/finetune/finetune_flux_grpo_MixGRPO.sh:120-196; paths swapped, a tiny random MMDiT saved
in the checkpoint format, a three-prompt embedding cache), two train steps with a checkpoint, then a second launch that
RESUMES from that checkpoint -- the loop of fastvideo/train_grpo_flux.py:803-853 (window scheduler from the flags,
LatentDataset + DistributedSampler(seed=sampler_seed), set_seed(seed + rank), checkpointing_steps -> save_checkpoint)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _flags(tmp, extra, epochs=1):
    base = f"""--seed 714 --pretrained_model_name_or_path {tmp}/flux --vae_model_path {tmp}/flux --cache_dir {tmp}/.cache
 --data_json_path {tmp}/rl_embeddings/videos2caption.json --gradient_checkpointing --train_batch_size 1 --num_latent_t 1
 --sp_size 1 --train_sp_batch_size 1 --dataloader_num_workers 0 --gradient_accumulation_steps 2
 --learning_rate 1e-4 --mixed_precision bf16 --allow_tf32 --cfg 0.0 --output_dir {tmp}/outputs
 --h 128 --w 128 --t 1 --sampling_steps 6 --eta 0.7 --lr_warmup_steps 0 --sampler_seed 1223627 --max_grad_norm 1.0
 --weight_decay 0.0001 --num_generations 4 --shift 3 --use_group --ignore_last --timestep_fraction 0.6 --init_same_noise
 --clip_range 1e-4 --adv_clip_max 5.0 --training_strategy part --experiment_name cli --kl_coeff 0.0
 --iters_per_group 1 --group_size 2 --sample_strategy progressive --prog_overlap --prog_overlap_step 1
 --max_iters_per_group 10 --min_iters_per_group 1 --roll_back --trimmed_ratio 0.0 --reward_model multi_reward
 --hps_path x --hps_clip_path x --clip_score_path x --image_reward_path x --image_reward_med_config x
 --image_reward_http_proxy none --image_reward_https_proxy none --pick_score_http_proxy none --pick_score_https_proxy none
 --unified_reward_url none --unified_reward_default_question_type semantic --unified_reward_num_workers 1
 --multi_reward_mix advantage_aggr --hps_weight 1.0 --clip_score_weight 1.0 --image_reward_weight 1.0 --pick_score_weight 1.0
 --unified_reward_weight 1.0 --dpm_algorithm_type null --dpm_apply_strategy post --dpm_post_compress_ratio 0.4
 --dpm_solver_order 2 --dpm_solver_type midpoint --frozen_init_timesteps -1 --wandb_key none --flow_grpo_sampling
 --mgx_max_epochs {epochs}"""
    return base.split() + extra


def _run(flags, tmp):
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, "-m", "mixgrpo_amd.train_grpo_flux"] + flags, cwd=str(tmp), env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]


def _setup(tmp):
    from mixgrpo_amd.flux import FluxConfig, FluxTransformer2DModel
    cfg = FluxConfig(num_layers=1, num_single_layers=1, attention_head_dim=128, num_attention_heads=4, joint_attention_dim=64,
                     pooled_projection_dim=32)
    FluxTransformer2DModel(cfg, device="cuda").init_synthetic(seed=5, std=0.05).save_pretrained(os.path.join(tmp, "flux", "transformer"))
    root = os.path.join(tmp, "rl_embeddings")
    for sub in ("prompt_embed", "pooled_prompt_embeds", "text_ids"):
        os.makedirs(os.path.join(root, sub))
    entries = []
    g = torch.Generator().manual_seed(0)
    for i in range(3):
        torch.save((0.3 * torch.randn(8, 64, generator=g)).bfloat16(), os.path.join(root, "prompt_embed", f"{i}.pt"))
        torch.save(torch.randn(32, generator=g).bfloat16(), os.path.join(root, "pooled_prompt_embeds", f"{i}.pt"))
        torch.save(torch.zeros(8, 3), os.path.join(root, "text_ids", f"{i}.pt"))
        entries.append({"prompt_embed_path": f"{i}.pt", "pooled_prompt_embeds_path": f"{i}.pt", "text_ids": f"{i}.pt",
                        "caption": f"prompt {i}", "length": 1})
    with open(os.path.join(root, "videos2caption.json"), "w") as f:
        json.dump(entries, f)


def test_main_with_the_launchers_flags_then_resume(tmp_path):
    tmp = str(tmp_path)
    _setup(tmp)
    # uninterrupted: 3 steps, checkpoint written at the start of step 3 (weights after two steps)
    logs = _run(_flags(tmp, ["--max_train_steps", "3", "--checkpointing_steps", "3"]), tmp)
    assert [l["step"] for l in logs] == [1, 2, 3] and [l["global_step"] for l in logs] == [0, 1, 2]
    assert [l["timesteps_train"] for l in logs] == [[0, 1], [1, 2], [2, 3]]          # iters_per_group 1, overlap step 1
    run_dir = os.path.join(tmp, "outputs", "part_cli")
    ck = os.path.join(run_dir, "checkpoint-3-0")
    for name in ("diffusion_pytorch_model.safetensors", "config.json", "optimizer.safetensors", "trainer_state.json",
                 "rng_state_rank0.safetensors"):
        assert os.path.exists(os.path.join(ck, name)), name
    assert json.load(open(os.path.join(run_dir, "args.json")))["num_generations"] == 4
    assert json.load(open(os.path.join(ck, "trainer_state.json")))["global_step"] == 2
    assert all(k in logs[0] for k in ("train_loss", "policy_loss", "kl_loss", "clip_frac", "grad_norm", "learning_rate",
                                      "reward_SyntheticReward"))

    # resumed: step 3 again, from the checkpoint -- same window, same rollout noise, same numbers as the uninterrupted step 3
    again = _run(_flags(tmp, ["--max_train_steps", "3", "--checkpointing_steps", "3", "--resume_from_checkpoint", ck,
                              "--experiment_name", "cli_resumed"]), tmp)
    assert [l["step"] for l in again] == [3] and again[0]["global_step"] == 2
    assert again[0]["timesteps_train"] == logs[2]["timesteps_train"]
    for k in ("train_loss", "grad_norm", "clip_frac", "reward_SyntheticReward"):
        assert again[0][k] == pytest.approx(logs[2][k], rel=1e-6, abs=1e-9), k
    assert not os.path.exists(os.path.join(tmp, "outputs", "part_cli_resumed", "checkpoint-3-0"))   # not rewritten


def test_resume_from_a_checkpoint_of_a_later_epoch(tmp_path):
    """The shipped launcher's shape in small: `max_train_steps` steps per epoch, a checkpoint every `checkpointing_steps`, so
    `checkpoint-{step}-{epoch}` with epoch > 0 is the normal case.  Two epochs of two steps, checkpoints at every step; the
    run resumed from `checkpoint-2-1` (taken before step 2 of epoch 1: three train steps done) continues at epoch 1, step 2,
    global step 3, with the prompt, window and noise of the uninterrupted run -- and does not restart at epoch 0."""
    tmp = str(tmp_path)
    _setup(tmp)
    logs = _run(_flags(tmp, ["--max_train_steps", "2", "--checkpointing_steps", "1"], epochs=2), tmp)
    assert [(l["epoch"], l["step"], l["global_step"]) for l in logs] == [(0, 1, 0), (0, 2, 1), (1, 1, 2), (1, 2, 3)]
    run_dir = os.path.join(tmp, "outputs", "part_cli")
    ck = os.path.join(run_dir, "checkpoint-2-1")
    st = json.load(open(os.path.join(ck, "trainer_state.json")))
    assert (st["epoch"], st["global_step"], st["steps_done"]) == (1, 1, 3)
    again = _run(_flags(tmp, ["--max_train_steps", "2", "--checkpointing_steps", "1", "--resume_from_checkpoint", ck,
                              "--experiment_name", "cli_e1"], epochs=2), tmp)
    assert [(l["epoch"], l["step"], l["global_step"]) for l in again] == [(1, 2, 3)]
    assert again[0]["timesteps_train"] == logs[3]["timesteps_train"]
    for k in ("train_loss", "grad_norm", "clip_frac", "reward_SyntheticReward"):
        assert again[0][k] == pytest.approx(logs[3][k], rel=1e-6, abs=1e-9), k
    assert not any(n.startswith("checkpoint") for n in os.listdir(os.path.join(tmp, "outputs", "part_cli_e1")))


def test_main_builds_the_decode_stage_and_scores_with_plugged_in_reward_models(tmp_path):
    """The reference's main() builds `AutoencoderKL.from_pretrained(<pretrained>, subfolder="vae", torch_dtype=bf16)`
    (train_grpo_flux.py:697-701) and scores decoded images (:279-316).  Here: a tiny random-init VAE directory next to the
    transformer, a toy reward model named on the command line (`--mgx_reward_plugin module:factory`), one train step through
    the script -- the HIP decode runs inside the rollout and the per-class reward reaches the log -- and an `--lr_scheduler`
    name other than the launcher's (`cosine`) is honoured, an unknown one fails with the list."""
    from safetensors.torch import save_file
    from oracle import vae as OV
    tmp = str(tmp_path)
    _setup(tmp)
    vdir = os.path.join(tmp, "flux", "vae")
    os.makedirs(vdir)
    vkw = dict(block_out_channels=(64, 64), layers_per_block=1, sample_size=32)
    json.dump({"_class_name": "AutoencoderKL", "block_out_channels": [64, 64], "layers_per_block": 1, "sample_size": 32,
               "latent_channels": 16, "out_channels": 3, "norm_num_groups": 32, "scaling_factor": 0.3611, "shift_factor": 0.1159,
               "mid_block_add_attention": True}, open(os.path.join(vdir, "config.json"), "w"))
    save_file({k: v.bfloat16().contiguous() for k, v in OV.init_params(OV.VaeConfig(**vkw), seed=6).items()},
              os.path.join(vdir, "diffusion_pytorch_model.safetensors"))
    with open(os.path.join(tmp, "toy_rewards.py"), "w") as f:
        f.write("def make(args):\n"
                "    def brightness(images, prompts):\n"
                "        assert len(images) == len(prompts) and images[0].dim() == 3 and images[0].shape[0] == 3\n"
                "        return [float(im.float().mean()) for im in images]\n"
                "    def contrast(images, prompts):\n"
                "        return [float(im.float().std()) for im in images]\n"
                "    return {'ToyBrightness': brightness, 'ToyContrast': contrast}\n")
    env_path = tmp + os.pathsep + ROOT
    flags = _flags(tmp, ["--max_train_steps", "1", "--checkpointing_steps", "100", "--mgx_reward_plugin", "toy_rewards:make",
                         "--lr_scheduler", "cosine", "--lr_warmup_steps", "4"])
    env = dict(os.environ, PYTHONPATH=env_path)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, "-m", "mixgrpo_amd.train_grpo_flux"] + flags, cwd=tmp, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "VAE loaded" in r.stdout and "HIP VAE decode + ToyBrightness, ToyContrast" in r.stdout
    logs = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(logs) == 1
    for k in ("reward_ToyBrightness", "reward_ToyContrast"):
        assert k in logs[0] and logs[0][k] == logs[0][k] and abs(logs[0][k]) < 1e3
    assert logs[0]["reward_ToyContrast"] > 0
    assert logs[0]["learning_rate"] == pytest.approx(1e-4 * 0.5)          # two optimizer steps (G 4 / accum 2) into the cosine schedule's 4-step warm-up
    bad = subprocess.run([sys.executable, "-m", "mixgrpo_amd.train_grpo_flux"] +
                         _flags(tmp, ["--max_train_steps", "1", "--lr_scheduler", "piecewise_constant"]), cwd=tmp, env=env,
                         capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "constant_with_warmup" in bad.stderr and "cosine_with_restarts" in bad.stderr
