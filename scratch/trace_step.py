"""Per-kernel totals of the LAST (timed) train step of a `rocprofv3 --kernel-trace` run of bench.py: the window is the last
`ms_per_step` milliseconds before the final kernel's end.  usage: trace_step.py <kernel_trace.csv> <ms_per_step> [out.csv]"""
import csv, re, sys, collections
path, ms = sys.argv[1], float(sys.argv[2])
rows = []
with open(path) as f:
    for x in csv.DictReader(f):
        n = re.sub(r"^void ", "", x["Kernel_Name"]).replace("(anonymous namespace)::", "").replace("at::native::", "")
        n = n.split("(")[0] if not n.startswith("gemm_") else n.split("(")[0]
        rows.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), n[:80]))
end = max(r[1] for r in rows)
t0 = end - int(ms * 1e6)
sel = [r for r in rows if r[0] >= t0]
tot = collections.defaultdict(lambda: [0, 0])
for s, e, n in sel:
    tot[n][0] += e - s
    tot[n][1] += 1
busy = sum(v[0] for v in tot.values())
out = sorted(tot.items(), key=lambda kv: -kv[1][0])
lines = ["kernel,calls,total_s,avg_us,pct_of_kernel_time"]
for n, (t, c) in out:
    lines.append(f"\"{n}\",{c},{t / 1e9:.4f},{t / c / 1e3:.1f},{100.0 * t / busy:.2f}")
lines.append(f"\"TOTAL kernel time in the window of {ms:.1f} ms\",{sum(v[1] for v in tot.values())},{busy / 1e9:.4f},,100")
text = "\n".join(lines)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(text + "\n")
print("\n".join(lines[:45]))
print(lines[-1])
