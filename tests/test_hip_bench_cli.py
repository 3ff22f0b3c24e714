"""bench.py end to end on the small stand-in workload: the one-GPU line carries the contract's fields (roofline,
cpu_baseline), and the N = 2 launch (the driver's `python -m torch.distributed.run ... bench.py --gpus 2` form, here with
both ranks on cuda:0 over gloo) runs its collectives to completion and reports the whole-job rate.  Guards the
distributed path of the bench against rank-asymmetric code (a rank-0-only train step deadlocks the gradient all-reduce)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from helpers import free_port

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(out):
    rows = [l for l in out.splitlines() if l.startswith('{"metric"')]
    assert len(rows) == 1, out[-2000:]
    return json.loads(rows[0])


def test_bench_one_gpu_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, "bench.py", "--workload", "tiny_256_T8_W2_G4", "--steps", "2", "--warmup", "1"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["value"] > 0
    assert j["unit"] == "images/s" and j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert abs(j["value"] - 4 / (j["ms_per_step"] / 1e3)) / j["value"] < 1e-3          # G = 4 images per step
    rf, cb = j["roofline"], j["cpu_baseline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["achieved"] > 0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    assert j["lib_version"] >= 100                              # a release build of the library (diagnostic builds: < 0)
    assert j["config"]["rollout_solver_steps"] == 8 and j["mfma_frac_train_step_executed"] <= j["mfma_frac_train_step"]
    vd = j["vae_decode"]                                        # reported beside the metric, never inside it
    assert vd["ms_per_image"] > 0 and "not part of `value`" in vd["note"]


def test_bench_two_ranks_on_one_gpu():
    env = dict(os.environ, MGX_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = free_port()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2",
                        "--workload", "tiny_256_T8_W2_G4", "--steps", "2", "--warmup", "1"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 2 and j["config"]["parallelism"] == "dp2" and j["config"]["global_batch"] == 8
    assert abs(j["value"] - 8 / (j["ms_per_step"] / 1e3)) / j["value"] < 1e-3          # both ranks' images over the max time
    assert j["cpu_baseline"] is None                                                   # reported at N = 1 only
    assert j["last_step"]["loss"] == j["last_step"]["loss"]
    # the line says by itself what the collective library saw
    d = j["dist"]
    assert d["backend"] == "gloo" and d["world_size"] == 2 and d["rccl_version"] is None and d["grad_dtype"] in ("bf16", "fp32")
    assert d["overlap"] in (True, False) and j["config"]["env_switches"].get("MGX_DIST_BACKEND") == "gloo"


def test_bench_refuses_more_rccl_ranks_than_devices():
    """`--gpus N` over RCCL with fewer than N visible devices must fail fast with a clear message, before any collective
    (two RCCL ranks on one device hang in the communicator set-up)."""
    env = {k: v for k, v in os.environ.items() if k not in ("MGX_DIST_BACKEND",)}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), "bench.py", "--gpus", "2",
                        "--workload", "tiny_256_T8_W2_G4", "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the devices")
    assert r.returncode != 0 and "needs 2 visible devices" in (r.stderr + r.stdout)


def test_bench_launches_its_own_ranks():
    """The driver's launcher-less form `python3 bench.py --gpus N ...` for N > 1 (no WORLD_SIZE): the parent starts the N
    ranks as child processes before touching the GPU, relays rank 0's JSON line and exits with the launcher's code."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MGX_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--workload", "tiny_256_T8_W2_G4", "--steps", "1",
                        "--warmup", "1", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 2 and j["config"]["parallelism"] == "dp2" and j["config"]["global_batch"] == 8
    assert j["roofline"]["achieved"] > 0                       # the profiled step is the (only) warm-up step
    # a failing launch is reported through the parent's exit code (here: 3 ranks asked, --gpus 2 given to each)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr",
                        "127.0.0.1", "--master-port", str(free_port()), "bench.py", "--gpus", "2", "--workload",
                        "tiny_256_T8_W2_G4"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout)


def test_bench_fp8_attention_switch_is_labelled():
    """`--attention fp8` (BASELINE.json configs[4]'s attention path) runs the same step and says so in `dtype`."""
    r = subprocess.run([sys.executable, "bench.py", "--workload", "tiny_256_T8_W2_G4", "--steps", "1", "--warmup", "1",
                        "--attention", "fp8", "--no-roofline", "--no-cpu-baseline"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _line(r.stdout)
    assert j["dtype"] == "bf16+fp8attn" and j["value"] > 0 and j["roofline"] is None and j["cpu_baseline"] is None
    assert j["last_step"]["loss"] == j["last_step"]["loss"]
