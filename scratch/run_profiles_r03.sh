#!/bin/bash
# Round-3 evidence run on the GPU box (gpurun): PMC passes (separate runs, --kernel-trace only besides --pmc) on the attention
# kernels (generated 64-wide forward / backward) and on the dominant GEMM; summaries are copied into profiles/ afterwards.
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r03_prof
mkdir -p $OUT
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  echo "[prof] attention pmc $tag"
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/attn_$tag -- python3 scratch/attn_one.py > $OUT/attn_$tag.log 2>&1 || exit 1
done
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  echo "[prof] gemm pmc $tag"
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/gemm_$tag -- python3 scratch/gemm_one.py 36864 3072 15360 2 > $OUT/gemm_$tag.log 2>&1 || exit 1
done
python3 scratch/pmc_summary.py $OUT/attn_pmc_summary.json $OUT/attn_SQ_VALU_MFMA_BUSY_CYCLES $OUT/attn_SQ_LDS_BANK_CONFLICT $OUT/attn_FETCH_SIZE $OUT/attn_WRITE_SIZE > /dev/null
python3 scratch/pmc_summary.py $OUT/gemm_pmc_summary.json $OUT/gemm_SQ_VALU_MFMA_BUSY_CYCLES $OUT/gemm_FETCH_SIZE $OUT/gemm_WRITE_SIZE > /dev/null
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
cat $OUT/attn_pmc_summary.json
