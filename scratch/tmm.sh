cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tmm -- python scratch/torch_mm.py > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/tmm/*/*kernel_trace.csv")[0]
seen=set()
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if n in seen: continue
    seen.add(n)
    print(n[:600]); print("LDS", r.get("LDS_Block_Size"), "VGPR", r.get("VGPR_Count"), "AGPR", r.get("Accum_VGPR_Count"), "WG", r.get("Workgroup_Size"), "grid", r.get("Grid_Size"), "dur_us", (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
PY
