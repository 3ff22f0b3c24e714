"""Derived figures from the PMC medians of scratch/pmc_summary.py (profiles/r02_*_pmc.json): effective clock
(GRBM_GUI_ACTIVE / 8 XCDs / wall), matrix-pipe busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (clock cycles x 1024 SIMDs)),
parked / issue-stalled wave fractions, and fabric traffic (FETCH_SIZE KiB doubled on gfx950, WRITE_SIZE KiB as is;
MI355X_MICROARCH.md section HBM).  usage: pmc_derive.py summary.json out.json algorithmic_bytes.json"""
import json, sys
src, out = sys.argv[1], sys.argv[2]
alg = json.loads(sys.argv[3]) if len(sys.argv) > 3 else {}
S = json.load(open(src))
res = {}
for k, v in S.items():
    if "GRBM_GUI_ACTIVE" not in v or not v.get("duration_ms"):
        continue
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0
    d = {"duration_ms_profiled_median": round(v["duration_ms"], 4), "dispatches_seen": v.get("dispatches_seen"),
         "effective_clock_ghz": round(cyc / (v["duration_ms"] * 1e-3) / 1e9, 3),
         "mfma_busy_frac": round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024), 3),
         "wave_wait_frac": round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 3),
         "wave_issue_stall_frac": round(v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 3)}
    if "SQ_LDS_IDX_ACTIVE" in v and v["SQ_LDS_IDX_ACTIVE"]:
        d["lds_conflict_frac_of_lds_cycles"] = round(v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"], 3)
    if "FETCH_SIZE" in v:
        rd, wr = v["FETCH_SIZE"] * 1024 * 2, v.get("WRITE_SIZE", 0.0) * 1024
        d.update(FETCH_SIZE_KiB_median=v["FETCH_SIZE"], WRITE_SIZE_KiB_median=v.get("WRITE_SIZE"),
                 hbm_read_bytes_corrected=rd, hbm_write_bytes=wr, traffic_bytes_per_launch=rd + wr)
        a = alg.get(k)
        if a:
            d["algorithmic_bytes_per_launch"] = a
            d["traffic_over_algorithmic"] = round((rd + wr) / a, 2)
    res[k] = d
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
