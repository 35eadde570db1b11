#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's roofline numbers on the GPU box.
#   usage (from the repo root, on the GPU box):  bash profiles/collect.sh <tag>
# Writes under gpurun_out/prof_<tag>/ and leaves the summaries to copy into profiles/:
#   <tag>_c3_fast_kernel_stats.csv   timeout -k 10 900 rocprofv3 --kernel-trace --stats of the bench command
#   <tag>_pmc/{sq_counters,fetch_size,write_size}.csv   separate --pmc passes (filtered to
#                                      this library's kernels), as MI355X_MICROARCH.md
#                                      section HBM prescribes
#   <tag>_pmc_summary.json           profiles/summarize_pmc.py over the three passes
set -e -o pipefail
TAG=${1:-r06}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT/pmc
export TMPDIR=/tmp
# no warm-up: the trace then holds exactly the kernels of the timed region (plus the one-off
# load kernels, which have names of their own), so its averages can be set against bench.py's.
# --serial_kernels: the backward sweep and est_maf between the objective rounds on the one
# stream (round 5 runs them NEXT TO the rounds on a second stream, where a kernel's span holds its
# neighbours' work too): every duration and every counter below is then the kernel's own.
BENCH="bench.py --steps 7 --warmup 0 --no_cpu_baseline --no_exact_line --no_check --no_cold --serial_kernels"

timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $BENCH \
  > $OUT/bench_under_trace.json 2> $OUT/trace.err
cp "$(find $OUT/trace -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_c3_fast_kernel_stats.csv
cp $OUT/bench_under_trace.json $OUT/${TAG}_bench_under_trace.json
python3 profiles/reconcile.py $OUT/${TAG}_c3_fast_kernel_stats.csv $OUT/${TAG}_bench_under_trace.json \
  > $OUT/${TAG}_trace_vs_bench.json
echo "trace done"

PMCBENCH="bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_exact_line --no_check --no_cold --serial_kernels"
filter() {  # keep the header and this library's kernels
  python3 - "$1" "$2" <<'EOF'
import csv, sys
src, dst = sys.argv[1:3]
with open(src) as f, open(dst, "w", newline="") as g:
    r = csv.reader(f); w = csv.writer(g)
    hdr = next(r); w.writerow(hdr)
    k = hdr.index("Kernel_Name")
    for row in r:
        if "nghmm" in row[k]:
            w.writerow(row)
EOF
}
timeout -k 10 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU \
  GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -o sq -- python3 $PMCBENCH \
  > /dev/null 2> $OUT/sq.err
filter "$(find $OUT/sq -name '*counter_collection.csv' | head -1)" $OUT/pmc/sq_counters.csv
echo "sq pass done"
timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 $PMCBENCH \
  > /dev/null 2> $OUT/fetch.err
filter "$(find $OUT/fetch -name '*counter_collection.csv' | head -1)" $OUT/pmc/fetch_size.csv
echo "fetch pass done"
timeout -k 10 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 $PMCBENCH \
  > /dev/null 2> $OUT/write.err
filter "$(find $OUT/write -name '*counter_collection.csv' | head -1)" $OUT/pmc/write_size.csv
echo "write pass done"
python3 profiles/summarize_pmc.py $OUT/pmc/sq_counters.csv $OUT/pmc/fetch_size.csv \
  $OUT/pmc/write_size.csv $OUT/${TAG}_pmc_summary.json > $OUT/summary.txt
# raw rocprof trees are large: keep only the summaries
rm -rf $OUT/trace $OUT/sq $OUT/fetch $OUT/write
ls -la $OUT $OUT/pmc
