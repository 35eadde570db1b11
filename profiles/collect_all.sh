#!/bin/bash
# One session on the GPU box = every number of the round from the same build:
#   bash profiles/collect_all.sh r03
# kernel trace + PMC passes of the default workload (profiles/collect.sh), then the bench lines:
# default (1000 x 1M), one rank's compute of the 2 / 4 / 8-GPU strong-scaling points (both shardings),
# configs[1] (100 x 100k) alone and with replicas, config 5's per-GPU share.
set -e -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
bash profiles/collect.sh $TAG
# bench.py reads profiles/<tag>_pmc_summary.json for `traffic`: install this session's first
cp $OUT/${TAG}_pmc_summary.json profiles/${TAG}_pmc_summary.json
python3 bench.py > $OUT/${TAG}_bench_default_n1.json
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
# config 5's share as a site shard: all 5000 individuals x 625 000 sites (one of eight ranks)
python3 bench.py --workload c5 --emulate_ranks 8 --no_cpu_baseline > $OUT/${TAG}_bench_c5_rank_of_8_sites.json
ls -la $OUT
