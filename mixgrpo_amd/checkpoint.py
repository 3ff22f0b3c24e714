"""Checkpoint format of the reference trainer + the resume it lacks.

`save_checkpoint(transformer, rank, output_dir, step, epoch)` mirrors fastvideo/utils/checkpoint.py:65-88: rank 0 writes
`{output_dir}/checkpoint-{step}-{epoch}/diffusion_pytorch_model.safetensors` (diffusers key names, fp32 master weights)
and `config.json` (the transformer's config without `dtype`), which `fastvideo/sample/sample_flux.py:28-30,464-465`
loads with `load_state_dict(strict=True)`.  Under replica data parallelism every rank holds the full weights, so the
reference's FSDP full-state-dict gather (:67-72) has no counterpart: rank 0 writes its own copy.

`save_resume_state` / `load_resume_state` add what the reference cannot do (train_grpo_flux.py:781-783 only parses
a step number from `--resume_from_checkpoint` and never restores anything): the AdamW moments and step count, the LR
schedule position, the SDE-window scheduler (`GRPOTrainingStates`) and the global step, next to the weights.
"""
import dataclasses
import json
import os

import torch

from .dist_utils import main_print


def save_checkpoint(transformer, rank, output_dir, step, epoch):
    main_print(f"--> saving checkpoint at step {step}")
    save_dir = os.path.join(output_dir, f"checkpoint-{step}-{epoch}")
    if rank <= 0:
        transformer.save_pretrained(save_dir)
    main_print(f"--> checkpoint saved at step {step}")
    return save_dir


def save_resume_state(save_dir, optimizer, lr_scheduler=None, grpo_states=None, global_step=0, rank=0, epoch=0,
                      steps_done=None):
    """Writes `optimizer.safetensors` (m, v: flat fp32 in the parameter store's order) and `trainer_state.json`.
    `global_step`: train steps finished inside the current epoch (the reference's inner loop position, `step - 1`);
    `epoch`: the outer loop's epoch (the `-{epoch}` of the checkpoint's directory name); `steps_done`: train steps finished
    over all epochs = prompt batches consumed = the reference's `global_step` counter (default: `global_step`, epoch 0)."""
    if rank > 0:
        return
    from safetensors.torch import save_file
    os.makedirs(save_dir, exist_ok=True)
    sd = optimizer.state_dict()
    state = {"global_step": int(global_step), "epoch": int(epoch),
             "steps_done": int(global_step if steps_done is None else steps_done),
             "optimizer_step": int(sd["step"]), "lr": float(sd["lr"]), "base_lr": float(optimizer.base_lr)}
    if lr_scheduler is not None:
        state["lr_scheduler"] = {"n": int(lr_scheduler.n), "warmup": int(lr_scheduler.warmup),
                                 "base": [float(b) for b in lr_scheduler.base]}
    if grpo_states is not None:
        state["grpo_states"] = dataclasses.asdict(grpo_states)
        state["grpo_states"]["init_timestep"] = int(grpo_states.init_timestep)
    # serialise the host state FIRST (a value JSON cannot take must not leave a half-written directory behind), write both
    # files under temporary names and rename them; trainer_state.json goes last: its presence marks a complete resume state
    text = json.dumps(state, indent=2, default=_json_scalar)
    tmp_t, tmp_j = os.path.join(save_dir, "optimizer.safetensors.tmp"), os.path.join(save_dir, "trainer_state.json.tmp")
    save_file({"m": sd["m"].detach().cpu(), "v": sd["v"].detach().cpu()}, tmp_t)
    with open(tmp_j, "w") as f:
        f.write(text)
    os.replace(tmp_t, os.path.join(save_dir, "optimizer.safetensors"))
    os.replace(tmp_j, os.path.join(save_dir, "trainer_state.json"))


def save_rng_state(save_dir, rank=0):
    """Every rank's generator states (python, numpy, torch CPU, torch device) next to the resume state, so that a resumed run
    draws the rollout noise the uninterrupted run would have drawn.  Plain tensors + JSON: loaded without unpickling."""
    import random

    import numpy as np
    from safetensors.torch import save_file
    os.makedirs(save_dir, exist_ok=True)
    tensors = {"torch_cpu": torch.get_rng_state()}
    if torch.cuda.is_available():
        tensors["torch_cuda"] = torch.cuda.get_rng_state()
    np_state = np.random.get_state()
    tensors["numpy_keys"] = torch.from_numpy(np_state[1].astype("int64"))
    py = random.getstate()
    meta = {"python": [py[0], list(py[1]), py[2]], "numpy": [np_state[0], int(np_state[2]), int(np_state[3]), float(np_state[4])]}
    save_file(tensors, os.path.join(save_dir, f"rng_state_rank{rank}.safetensors"))
    with open(os.path.join(save_dir, f"rng_state_rank{rank}.json"), "w") as f:
        json.dump(meta, f)


def load_rng_state(save_dir, rank=0):
    """Restores what `save_rng_state` wrote; returns False when the directory holds none for this rank."""
    import random

    import numpy as np
    from safetensors.torch import load_file
    path = os.path.join(save_dir, f"rng_state_rank{rank}.safetensors")
    if not os.path.exists(path):
        return False
    tensors = load_file(path)
    with open(os.path.join(save_dir, f"rng_state_rank{rank}.json")) as f:
        meta = json.load(f)
    torch.set_rng_state(tensors["torch_cpu"])
    if "torch_cuda" in tensors and torch.cuda.is_available():
        torch.cuda.set_rng_state(tensors["torch_cuda"])
    n = meta["numpy"]
    np.random.set_state((n[0], tensors["numpy_keys"].numpy().astype("uint32"), n[1], n[2], n[3]))
    py = meta["python"]
    random.setstate((py[0], tuple(py[1]), py[2]))
    return True


def _json_scalar(x):
    """numpy scalars (the exp_decay budget is a numpy float, window indices may be numpy ints) as python numbers."""
    if hasattr(x, "item"):
        return x.item()
    raise TypeError(f"{type(x).__name__} is not JSON serialisable")


def load_resume_position(save_dir):
    """(epoch, steps finished inside that epoch, steps finished over all epochs) of a resume state; states written before
    the epoch was recorded read as epoch 0."""
    with open(os.path.join(save_dir, "trainer_state.json")) as f:
        state = json.load(f)
    return int(state.get("epoch", 0)), int(state["global_step"]), int(state.get("steps_done", state["global_step"]))


def load_resume_state(save_dir, optimizer, lr_scheduler=None, grpo_states=None):
    """Restores what `save_resume_state` wrote (weights are loaded separately with `from_pretrained` /
    `load_state_dict`).  Returns the number of steps finished inside the saved epoch (`load_resume_position` has the epoch)."""
    from safetensors.torch import load_file
    with open(os.path.join(save_dir, "trainer_state.json")) as f:
        state = json.load(f)
    mv = load_file(os.path.join(save_dir, "optimizer.safetensors"))
    if mv["m"].numel() != optimizer.m.numel():
        raise ValueError(f"optimizer state has {mv['m'].numel()} elements, the model's flat store {optimizer.m.numel()}")
    optimizer.load_state_dict({"step": state["optimizer_step"], "m": mv["m"], "v": mv["v"], "lr": state["lr"]})
    if lr_scheduler is not None and "lr_scheduler" in state:
        lr_scheduler.n = state["lr_scheduler"]["n"]
        lr_scheduler.warmup = state["lr_scheduler"]["warmup"]
        lr_scheduler.base = list(state["lr_scheduler"]["base"])
        lr_scheduler._apply()
    if grpo_states is not None and "grpo_states" in state:
        for k, v in state["grpo_states"].items():
            setattr(grpo_states, k, v)
    return state["global_step"]
