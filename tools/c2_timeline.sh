#!/bin/bash
# Kernel timeline of configs[1] (100 x 100k): rocprofv3 --kernel-trace of a short bench run, then
# tools/c2_timeline.py prints one steady-state EM iteration (kernel, queue, start offset, duration).
#   usage (repo root, GPU box):  bash tools/c2_timeline.sh [workload] [extra bench args]
set -e -o pipefail
WL=${1:-c2}; shift || true
OUT=gpurun_out/timeline_$WL
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 bench.py \
  --workload $WL --steps 10 --warmup 6 --no_cpu_baseline --no_exact_line --no_check --no_cold "$@" \
  > $OUT/bench.json 2> $OUT/trace.err
cp "$(find $OUT/trace -name '*kernel_trace.csv' | head -1)" $OUT/kernel_trace.csv
rm -rf $OUT/trace
python3 tools/c2_timeline.py $OUT/kernel_trace.csv > $OUT/timeline.txt
tail -n 80 $OUT/timeline.txt
