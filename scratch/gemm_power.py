"""Board power and shader clock sampled beside a steady loop of the dominant GEMM (M 36864, N 3072, K 15360, gate-residual
epilogue) for several tile orders (`MGX_GEMM_BAND`, one process each: the variable is read once): does the order that moves
fewer bytes beyond L2 buy clock?  Sampling: `rocm-smi --showpower --showclocks --json` every 0.5 s from a thread."""
import json, os, subprocess, sys, threading, time

def child(band):
    import torch
    sys.path.insert(0, ".")
    from mixgrpo_amd import ops
    from mixgrpo_amd.ops import Rows
    torch.manual_seed(0)
    M, N, K = 36864, 3072, 15360
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16(); C = torch.randn(M, N, device="cuda").bfloat16()
    gate = torch.randn(8, N, device="cuda").bfloat16()
    def run():
        ops.gemm(Rows(A, M, K, M // 8, (M // 8) * K), W, b, Rows(C, M, N, M // 8, (M // 8) * N), N, K, 2, gate=gate, gate_ld=N)
    for _ in range(20): run()
    torch.cuda.synchronize()
    samples, stop = [], False
    def poll():
        while not stop:
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
                samples.append(json.loads(out))
            except Exception as e:
                samples.append({"error": str(e)})
            time.sleep(0.5)
    th = threading.Thread(target=poll); th.start()
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 6.0:
        for _ in range(50): run()
        n += 50
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop = True; th.join()
    ms = e0.elapsed_time(e1) / n
    def grab(s, key):
        vals = []
        for card in s.values() if isinstance(s, dict) else []:
            if isinstance(card, dict):
                for k, v in card.items():
                    if key in k.lower():
                        try: vals.append(float(str(v).strip("()Mhz W").split()[0].replace("Mhz", "")))
                        except Exception: pass
        return vals
    pw = [x for s in samples for x in grab(s, "power")]
    sc = [x for s in samples for x in grab(s, "sclk")]
    print(json.dumps({"band": band, "ms_per_gemm": round(ms, 4), "tflops": round(2.0 * M * N * K / ms / 1e9, 1), "samples": len(samples),
                      "power_w_median": (sorted(pw)[len(pw) // 2] if pw else None), "power_w_max": (max(pw) if pw else None),
                      "sclk_mhz_median": (sorted(sc)[len(sc) // 2] if sc else None), "raw_first": samples[0] if samples else None}), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for band in ("0", "1", "4", "12"):
            env = dict(os.environ)
            if band != "0":
                env["MGX_GEMM_BAND"] = band
            r = subprocess.run([sys.executable, __file__, band], env=env, capture_output=True, text=True, timeout=120)
            print(r.stdout.strip()[-1500:] or r.stderr[-800:], flush=True)
