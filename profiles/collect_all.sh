#!/bin/bash
# One session on the GPU box = every number of the round from the same build:
#   bash profiles/collect_all.sh r05
# kernel trace + PMC passes of the default workload (profiles/collect.sh), then the bench lines:
# default (1000 x 1M), one rank's compute of the 2 / 4 / 8-GPU strong-scaling points (both shardings),
# configs[1] (100 x 100k) alone and with replicas, config 5's per-GPU share.
# (a gpurun call is at most 20 minutes: `collect_all.sh r04 1` = the profiler passes and the default
# lines, `collect_all.sh r04 2` = the other workloads; the committed profiles/<tag>_pmc_summary.json of
# part 1 must be in place for part 2's lines to carry `traffic`)
set -e -o pipefail
TAG=${1:-r06}
PART=${2:-all}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
if [ "$PART" != 2 ]; then
bash profiles/collect.sh $TAG
# bench.py reads profiles/<tag>_pmc_summary.json for `traffic`: install this session's first
cp $OUT/${TAG}_pmc_summary.json profiles/${TAG}_pmc_summary.json
python3 bench.py > $OUT/${TAG}_bench_default_n1.json
# the one-GPU result check every N > 1 line compares itself with (profiles/check_n1.json)
python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_exact_line --write_check > /dev/null
cp profiles/check_n1.json $OUT/check_n1.json
fi
if [ "$PART" = 1 ]; then ls -la $OUT; exit 0; fi
# one rank's compute of the 2 / 4 / 8-GPU strong-scaling points (exchanges as local copies):
# site shards (fast mode's default) and individual shards
# (the driver's N = 1 flags, --steps 20 --warmup 5: the first iterations of a run take more
# objective rounds than the steady state, and a 10-iteration window right after 3 warm-up ones
# still catches some of them)
W="--steps 20 --warmup 5 --no_cpu_baseline"
python3 bench.py $W > $OUT/${TAG}_bench_n1_steps20.json
for v in 2 4 8; do
  python3 bench.py $W --emulate_ranks $v > $OUT/${TAG}_bench_rank_of_${v}_sites.json
  python3 bench.py $W --emulate_ranks $v --shard individuals > $OUT/${TAG}_bench_rank_of_${v}_individuals.json
  # ... the site shards' all-gathers through a one-rank RCCL group on the handle's stream
  python3 bench.py $W --emulate_ranks $v --emulate_rccl > $OUT/${TAG}_bench_rank_of_${v}_sites_rccl_in_loop.json
done
python3 bench.py --workload c2 --no_cpu_baseline --steps 200 --warmup 20 > $OUT/${TAG}_bench_c2_n1.json
python3 bench.py --workload c2 --no_cpu_baseline --steps 100 --warmup 10 --replicas 10 > $OUT/${TAG}_bench_c2_replicas10_n1.json
python3 bench.py --workload c5share --no_cpu_baseline > $OUT/${TAG}_bench_c5share_n1.json
# off examples/test.sh's operating point (round-5 review, item 1): the simulator's `r` options at depth 5
# and 2, a transition rate beyond the small-alpha kernels, likelihood cohorts above 1024 individuals; the
# driver's flags; the CPU restatement beside the first of them
python3 bench.py --workload c3r --steps 20 --warmup 5 --no_exact_line > $OUT/${TAG}_bench_c3r_n1.json
for w in c3r2 c3hi c2k c5k; do
  python3 bench.py --workload $w --steps 20 --warmup 5 --no_cpu_baseline --no_exact_line > $OUT/${TAG}_bench_${w}_n1.json
done
python3 bench.py --workload c2r --no_cpu_baseline --no_exact_line --steps 200 --warmup 20 > $OUT/${TAG}_bench_c2r_n1.json
# one steady-state iteration of configs[1] as the device ran it (tools/c2_timeline.py)
bash tools/c2_timeline.sh c2 > /dev/null && cp gpurun_out/timeline_c2/timeline.txt $OUT/${TAG}_c2_timeline.txt
# config 5's share as a site shard: all 5000 individuals x 625 000 sites (one of eight ranks)
python3 bench.py --workload c5 --emulate_ranks 8 --no_cpu_baseline > $OUT/${TAG}_bench_c5_rank_of_8_sites.json
# the called genotypes' est_maf (k_fast_estmaf_called_sums: one sweep over codes and posteriors) on
# config 5's rank: kernel trace + FETCH_SIZE / WRITE_SIZE passes of a two-iteration run
C5="bench.py --workload c5 --emulate_ranks 8 --steps 2 --warmup 1 --no_cpu_baseline --no_cold"
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5trace -o trace -- python3 $C5 > /dev/null 2> $OUT/c5trace.err
cp "$(find $OUT/c5trace -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_c5_rank_kernel_stats.csv
timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c5fetch -o fetch -- python3 $C5 > /dev/null 2> $OUT/c5fetch.err
timeout -k 10 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/c5write -o write -- python3 $C5 > /dev/null 2> $OUT/c5write.err
python3 profiles/summarize_called.py $OUT/${TAG}_c5_rank_kernel_stats.csv \
  "$(find $OUT/c5fetch -name '*counter_collection.csv' | head -1)" \
  "$(find $OUT/c5write -name '*counter_collection.csv' | head -1)" 5000 625000 > $OUT/${TAG}_c5_estmaf_called.json
rm -rf $OUT/c5trace $OUT/c5fetch $OUT/c5write
ls -la $OUT
