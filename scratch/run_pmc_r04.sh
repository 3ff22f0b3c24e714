#!/bin/bash
# Round-4 PMC passes on the GEMM (separate runs per counter group, --kernel-trace only besides --pmc): the dominant shape
# (MODE 0: the round-3 tile walk) and the 2.25-round weight-gradient shape with the stream-K tail (MODE 1 + the fix-up kernel).
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r04_pmc
mkdir -p $OUT
for shape in "36864 3072 15360 2" "3072 12288 16384 3"; do
  tag=$(echo $shape | tr ' ' '_')
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
    g=$(echo $grp | cut -d' ' -f1)
    echo "[prof] gemm $tag pmc $g"
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/${tag}_$g -- python3 scratch/gemm_one.py $shape > $OUT/${tag}_$g.log 2>&1 || exit 1
  done
  python3 scratch/pmc_summary.py $OUT/summary_$tag.json $OUT/${tag}_SQ_VALU_MFMA_BUSY_CYCLES $OUT/${tag}_FETCH_SIZE $OUT/${tag}_WRITE_SIZE > /dev/null
  python3 scratch/pmc_derive.py $OUT/summary_$tag.json $OUT/derived_$tag.json > /dev/null
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
cat $OUT/derived_*.json
