"""Summarise rocprofv3 --pmc passes (one directory per counter group) into one JSON: per kernel, the median of every
counter and of the dispatch duration.  usage: pmc_summary.py out.json dir1 dir2 ..."""
import csv, glob, json, statistics, sys
out, dirs = sys.argv[1], sys.argv[2:]
acc = {}
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = {}
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            key = (name, r["Dispatch_Id"])
            e = rows.setdefault(key, {"dur": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 if r.get("End_Timestamp") else None})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for (name, _), e in rows.items():
            a = acc.setdefault(name, {})
            for k, v in e.items():
                if v is not None:
                    a.setdefault(k if k != "dur" else "duration_ms", []).append(v)
res = {}
for name, a in acc.items():
    if not any(s in name for s in ("attn", "fp8", "flow_", "dance_", "dpm_", "logp", "gemm_persist", "gemm_pp", "gemm_sk")):
        continue
    res[name] = {k: statistics.median(v) for k, v in a.items()}
    res[name]["dispatches_seen"] = max(len(v) for v in a.values())
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
