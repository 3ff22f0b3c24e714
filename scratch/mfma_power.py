"""Scratch experiment: sustained bf16 MFMA rate of the two instruction shapes (no memory traffic at all) under the board's
power cap -- does 32x32x16 (half the operand-register reads per FLOP) hold a higher clock than 16x16x32?"""
import ctypes, json, os, subprocess, sys, threading, time

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mfma_power.so"))
lib.run.restype = ctypes.c_double
lib.run.argtypes = [ctypes.c_int] * 5

def grab(s, key):
    vals = []
    for card in s.values() if isinstance(s, dict) else []:
        if isinstance(card, dict):
            for k, v in card.items():
                if key in k.lower():
                    try: vals.append(float(str(v).strip("()Mhz W").split()[0].replace("Mhz", "")))
                    except Exception: pass
    return vals

def measure(variant, zero, wps, seconds=6.0):
    lib.run(variant, 20000, zero, 2, wps)
    samples, stop = [], [False]
    def poll():
        while not stop[0]:
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
                samples.append(json.loads(out))
            except Exception as e:
                samples.append({"error": str(e)})
            time.sleep(0.5)
    th = threading.Thread(target=poll); th.start()
    t0 = time.time(); rates = []
    while time.time() - t0 < seconds:
        rates.append(lib.run(variant, 200000, zero, 4, wps))
    stop[0] = True; th.join()
    pw = sorted(x for s in samples for x in grab(s, "power")); sc = sorted(x for s in samples for x in grab(s, "sclk"))
    print(json.dumps({"mfma": "16x16x32" if variant == 16 else "32x32x16", "data": "zeros" if zero else "random", "waves_per_simd": wps,
                      "tflops_last": round(rates[-1], 1), "tflops_mean_2nd_half": round(sum(rates[len(rates) // 2:]) / len(rates[len(rates) // 2:]), 1),
                      "power_w_median": pw[len(pw) // 2] if pw else None, "sclk_mhz_median": sc[len(sc) // 2] if sc else None,
                      "n": len(rates)}), flush=True)

for wps in (1, 2):
    for zero in (0, 1):
        for variant in (16, 32):
            measure(variant, zero, wps)
