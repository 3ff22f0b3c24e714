#!/bin/bash
# Round-4 evidence run on the GPU box: kernel trace + stats of bench.py (2 warm-up steps + 1 timed), the LAST step cut out of the
# trace by scratch/trace_step.py, the per-shape GEMM table of the roofline step.  Summaries are copied into profiles/ afterwards.
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r04
TAG=${1:-a}
shift
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$TAG -- python3 bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-vae --gemm-shapes $OUT/gemm_shapes_$TAG.jsonl "$@" > $OUT/bench_trace_$TAG.log 2>&1 || exit 1
f=$(find $OUT/trace_$TAG -name "*kernel_trace.csv" | head -1)
ms=$(grep "^{\"metric\"" $OUT/bench_trace_$TAG.log | tail -1 | python3 -c "import json,sys;print(json.loads(sys.stdin.read())['ms_per_step'])")
python3 scratch/trace_step.py $f $ms $OUT/trainstep_kernel_stats_$TAG.csv
s=$(find $OUT/trace_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$s" ] && cp $s $OUT/rocprof_kernel_stats_$TAG.csv
rm -rf $OUT/trace_$TAG
grep "^{\"metric\"" $OUT/bench_trace_$TAG.log | tail -1 | cut -c1-200
