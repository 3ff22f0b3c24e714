"""Names the limiter behind the clocks the MFMA-only loop and the dominant GEMM hold: `amd-smi metric
--violation --power --clock --json` (MI300+: accumulated throttle counters per cause -- PPT = package power, socket / VR /
HBM thermal, PROCHOT -- and the percentage of the last window each was active) sampled every 0.5 s beside
  (a) back-to-back bf16 MFMAs on random register operands, two waves per SIMD, both instruction shapes (scratch/mfma_power.hip)
  (b) the dominant GEMM (M 36864, N 3072, K 15360, gate-residual epilogue) through the product library.
Prints one JSON line per load: rate, median power / clock, and for every violation field the first / last value seen."""
import ctypes, json, os, subprocess, sys, threading, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))


def smi():
    try:
        out = subprocess.run(["amd-smi", "metric", "-g", "0", "--violation", "--power", "--clock", "--json"], capture_output=True,
                             text=True, timeout=10).stdout
        return json.loads(out)
    except Exception as e:
        return {"error": str(e)}


def flat(d, pre=""):
    out = {}
    if isinstance(d, dict):
        for k, v in d.items():
            out.update(flat(v, f"{pre}{k}."))
    elif isinstance(d, list):
        for i, v in enumerate(d):
            out.update(flat(v, f"{pre}{i}."))
    else:
        out[pre[:-1]] = d
    return out


def sample_during(fn, seconds=6.0):
    samples, stop = [], [False]

    def poll():
        while not stop[0]:
            samples.append(flat(smi()))
            time.sleep(0.5)
    th = threading.Thread(target=poll)
    th.start()
    t0 = time.time()
    rates = []
    while time.time() - t0 < seconds:
        rates.append(fn())
    stop[0] = True
    th.join()
    keys = sorted({k for s in samples for k in s})
    summary = {}
    gfx = []
    for k in keys:
        vals = [s[k] for s in samples if k in s]
        nums = [v for v in vals if isinstance(v, (int, float)) and not isinstance(v, bool)]
        low = k.lower()
        short = k.replace("gpu_data.0.", "")
        if ".xcp_" in low or low.endswith(".unit"):
            continue
        if ".throttle." in low:                                   # accumulators / status / activity: first and last sample
            if vals[0] != vals[-1] or vals[-1] not in (0, "NOT ACTIVE", "N/A"):
                summary[short] = [vals[0], vals[-1]]
        elif low.endswith("socket_power.value") and nums:
            summary["socket_power_w_median"] = sorted(nums)[len(nums) // 2]
        elif ".clock.gfx_" in low and low.endswith(".clk.value") and nums:
            gfx.append(sorted(nums)[len(nums) // 2])
    if gfx:
        summary["gfx_clk_mhz_median_per_xcd"] = gfx
    return rates, summary, len(samples), (samples[len(samples) // 2] if samples else None)


def mfma_part():
    lib = ctypes.CDLL(os.path.join(HERE, "mfma_power.so"))
    lib.run.restype = ctypes.c_double
    lib.run.argtypes = [ctypes.c_int] * 5
    idle = flat(smi())
    print(json.dumps({"idle": {k.replace("gpu_data.0.", ""): v for k, v in idle.items()
                               if ".xcp_" not in k and (".throttle." in k or k.endswith("socket_power.value"))}}), flush=True)
    for variant in (16, 32):
        lib.run(variant, 20000, 0, 2, 2)
        rates, summ, n, _ = sample_during(lambda v=variant: lib.run(v, 200000, 0, 4, 2))
        half = rates[len(rates) // 2:]
        print(json.dumps({"load": f"mfma_only_{'16x16x32' if variant == 16 else '32x32x16'}_random_2waves",
                          "tflops_mean_2nd_half": round(sum(half) / max(1, len(half)), 1), "samples": n, "smi": summ}), flush=True)


def gemm_part():
    import torch
    from mixgrpo_amd import ops
    from mixgrpo_amd.ops import Rows
    torch.manual_seed(0)
    M, N, K = 36864, 3072, 15360
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    C = torch.randn(M, N, device="cuda").bfloat16()
    gate = torch.randn(8, N, device="cuda").bfloat16()

    def run():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.gemm(Rows(A, M, K, M // 8, (M // 8) * K), W, b, Rows(C, M, N, M // 8, (M // 8) * N), N, K, 2, gate=gate, gate_ld=N)
        e1.record()
        torch.cuda.synchronize()
        return 2.0 * M * N * K * 50 / e0.elapsed_time(e1) / 1e9
    run()
    rates, summ, n, _ = sample_during(run)
    half = rates[len(rates) // 2:]
    print(json.dumps({"load": "gemm_pp_kernel<2> M36864 N3072 K15360", "tflops_mean_2nd_half": round(sum(half) / max(1, len(half)), 1),
                      "samples": n, "smi": summ}), flush=True)


if __name__ == "__main__":
    # two processes: the MFMA loop's bare HIP library and torch do not share one (torch then finds no device)
    if len(sys.argv) > 1:
        (mfma_part if sys.argv[1] == "mfma" else gemm_part)()
    else:
        for part in ("mfma", "gemm"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), part])
            if r.returncode != 0:
                sys.exit(r.returncode)
