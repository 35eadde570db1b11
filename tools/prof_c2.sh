cd /tmp && export TMPDIR=/tmp
R=${1:-1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_c2_r$R
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --workload c2 --steps 20 --warmup 0 --no_cpu_baseline --replicas $R > $OUT/bench.json 2> $OUT/err.txt
cp "$(find $OUT/trace -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats.csv
rm -rf $OUT/trace
