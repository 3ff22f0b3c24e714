#!/bin/bash
# Round-2 evidence run on the GPU box (gpurun): kernel-trace stats of one train step + PMC passes (separate runs, no trace
# domains besides --kernel-trace: gpurun refuses other combinations) on the dominant GEMM and the attention kernels.
# Everything lands under gpurun_out/r02_prof/; the summaries to keep are copied into profiles/ afterwards.
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r02_prof
mkdir -p $OUT
echo "[prof] kernel-trace stats of warm-up + one timed train step"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trainstep -- python3 bench.py --steps 1 --warmup 1 --no-roofline --no-cpu-baseline > $OUT/trainstep_bench.log 2> $OUT/trainstep_bench.err || exit 1
tail -1 $OUT/trainstep_bench.log | cut -c1-300
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  echo "[prof] gemm pmc $tag"
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/gemm_$tag -- python3 scratch/gemm_one.py 36864 3072 15360 2 > $OUT/gemm_$tag.log 2>&1 || exit 1
done
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  echo "[prof] attention pmc $tag"
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/attn_$tag -- python3 scratch/attn_one.py > $OUT/attn_$tag.log 2>&1 || exit 1
done
python3 scratch/pmc_summary.py $OUT/gemm_pmc_summary.json $OUT/gemm_SQ_VALU_MFMA_BUSY_CYCLES $OUT/gemm_FETCH_SIZE $OUT/gemm_WRITE_SIZE > /dev/null
python3 scratch/pmc_summary.py $OUT/attn_pmc_summary.json $OUT/attn_SQ_VALU_MFMA_BUSY_CYCLES $OUT/attn_SQ_LDS_BANK_CONFLICT $OUT/attn_FETCH_SIZE $OUT/attn_WRITE_SIZE > /dev/null
find $OUT/trainstep -name "*kernel_stats.csv" -exec cp {} $OUT/trainstep_kernel_stats.csv \;
# the raw per-dispatch CSVs are large: keep the summaries only
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
du -sh $OUT; ls $OUT
