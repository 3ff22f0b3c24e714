"""The drop-in entry point's command line (`python -m mixgrpo_amd.train_grpo_flux`): it must take the flag list the
reference's launcher passes (scripts/finetune/finetune_flux_grpo_MixGRPO.sh:120-196) unchanged -- the flags the hot path
reads, the ones only main() reads, and the ones this fork defines but never reads -- with the
reference parser's defaults (fastvideo/train_grpo_flux.py:894-1423)."""
from mixgrpo_amd.train_grpo_flux import build_parser, reward_weights_from_args

# the launcher's flag list with its shell variables at the values the script sets (sh:40-80)
SCRIPT_FLAGS = """--seed 714 --pretrained_model_name_or_path ./data/flux --vae_model_path ./data/flux --cache_dir data/.cache
 --data_json_path data/rl_embeddings/videos2caption.json --gradient_checkpointing --train_batch_size 1 --num_latent_t 1
 --sp_size 1 --train_sp_batch_size 1 --dataloader_num_workers 4 --gradient_accumulation_steps 3 --max_train_steps 300
 --learning_rate 1e-5 --mixed_precision bf16 --checkpointing_steps 50 --allow_tf32 --cfg 0.0 --output_dir data/outputs
 --h 720 --w 720 --t 1 --sampling_steps 25 --eta 0.7 --lr_warmup_steps 0 --sampler_seed 1223627 --max_grad_norm 1.0
 --weight_decay 0.0001 --num_generations 12 --shift 3 --use_group --ignore_last --timestep_fraction 0.6 --init_same_noise
 --clip_range 1e-4 --adv_clip_max 5.0 --training_strategy part --experiment_name 0714_test --kl_coeff 0.0
 --iters_per_group 25 --group_size 4 --sample_strategy progressive --prog_overlap --prog_overlap_step 1
 --max_iters_per_group 10 --min_iters_per_group 1 --roll_back --trimmed_ratio 0.0 --reward_model multi_reward
 --hps_path ./hps_ckpt/HPS_v2.1_compressed.pt --hps_clip_path ./hps_ckpt/open_clip_pytorch_model.bin
 --clip_score_path ./clip_score_ckpt --image_reward_path ./image_reward_ckpt/ImageReward.pt
 --image_reward_med_config ./image_reward_ckpt/med_config.json --image_reward_http_proxy none --image_reward_https_proxy none
 --pick_score_http_proxy none --pick_score_https_proxy none --unified_reward_url none
 --unified_reward_default_question_type semantic --unified_reward_num_workers 1 --multi_reward_mix advantage_aggr
 --hps_weight 1.0 --clip_score_weight 1.0 --image_reward_weight 1.0 --pick_score_weight 1.0 --unified_reward_weight 1.0
 --dpm_algorithm_type null --dpm_apply_strategy post --dpm_post_compress_ratio 0.4 --dpm_solver_order 2
 --dpm_solver_type midpoint --frozen_init_timesteps -1 --wandb_key none --flow_grpo_sampling""".split()


def test_parser_takes_the_launchers_flag_list():
    a = build_parser().parse_args(SCRIPT_FLAGS)
    assert (a.h, a.w, a.t, a.sampling_steps, a.num_generations, a.shift, a.eta) == (720, 720, 1, 25, 12, 3.0, 0.7)
    assert a.use_group and a.init_same_noise and a.flow_grpo_sampling and a.prog_overlap and a.roll_back and a.ignore_last
    assert not a.drop_last_sample and a.training_strategy == "part" and a.dpm_solver_type == "midpoint"
    assert (a.gradient_accumulation_steps, a.max_grad_norm, a.clip_range, a.kl_coeff) == (3, 1.0, 1e-4, 0.0)
    assert reward_weights_from_args(a) == {"HPSClipRewardModel": 1.0, "ImageRewardModel": 1.0, "PickScoreRewardModel": 1.0}


def test_parser_defaults_are_the_reference_parsers():
    a = build_parser().parse_args(["--data_json_path", "x.json"])
    want = dict(dataloader_num_workers=10, train_batch_size=16, checkpointing_steps=500, gradient_accumulation_steps=1,
                learning_rate=1e-4, lr_warmup_steps=10, max_grad_norm=2.0, weight_decay=0.01, lr_scheduler="constant_with_warmup",
                master_weight_type="fp32", num_generations=16, shift=1.0, timestep_fraction=1.0, clip_range=1e-4, adv_clip_max=5.0,
                advantage_rerange_strategy="null", training_strategy="all", frozen_init_timesteps=-1, kl_coeff=0.01,
                iters_per_group=25, group_size=4, sample_strategy="progressive", prog_overlap_step=1, max_iters_per_group=10,
                min_iters_per_group=1, reward_model="hpsv2", multi_reward_mix="advantage_aggr", dpm_algorithm_type="null",
                dpm_apply_strategy="post", dpm_post_compress_ratio=0.4, dpm_solver_order=2, dpm_solver_type="heun",
                use_group=False, init_same_noise=False, flow_grpo_sampling=False, drop_last_sample=False, roll_back=False,
                h=None, w=None, t=None, sampling_steps=None, eta=None, seed=None, sampler_seed=None, output_dir=None,
                resume_from_checkpoint=None, max_train_steps=None, fsdp_sharding_startegy="full", selective_checkpointing=1.0)
    for k, v in want.items():
        assert getattr(a, k) == v, (k, getattr(a, k), v)
    assert len(vars(a)) >= 91                                   # the reference parser defines 91 flags
