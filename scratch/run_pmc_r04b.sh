#!/bin/bash
# Round-4 (second half) PMC passes on the rollout's new kernels: separate runs per counter group, --kernel-trace only besides --pmc.
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r04_pmc_b
mkdir -p $OUT
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  g=$(echo $grp | cut -d' ' -f1)
  echo "[prof] fused_one pmc $g"
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/$g -- python3 scratch/fused_one.py > $OUT/$g.log 2>&1 || exit 1
done
python3 scratch/pmc_summary.py $OUT/summary.json $OUT/SQ_VALU_MFMA_BUSY_CYCLES $OUT/FETCH_SIZE $OUT/WRITE_SIZE > /dev/null
python3 scratch/pmc_derive.py $OUT/summary.json $OUT/derived.json '{"attn_fwd64_kernel<true>": 905969664, "gemm_pp_kernel<6, false, 0>": 717225984, "gemm_pp_kernel<0, false, 0>": 471859200}' > /dev/null
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
cat $OUT/derived.json
