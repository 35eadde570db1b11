#!/bin/bash
# Exploration on the GPU box: the regime cases of tests/test_gpu_baseline_oracle.py, then one bench line per
# regime workload (no CPU baseline, no exact-mode line): bash tools/regimes_probe.sh <tag> [workloads...]
set -o pipefail
TAG=${1:-probe}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
WLS=${@:-c3 c3r c3r2 c3hi c2 c2r c2k c5k}
for w in $WLS; do
  echo "== $w" | tee -a $OUT/progress.log
  timeout -k 10 400 python3 bench.py --workload $w --steps 10 --warmup 5 --no_cpu_baseline --no_exact_line \
      > $OUT/bench_$w.json 2> $OUT/bench_$w.err || { echo "bench $w failed: $?" | tee -a $OUT/progress.log; tail -5 $OUT/bench_$w.err; }
  python3 - <<PY | tee -a $OUT/progress.log
import json
try:
    d = json.load(open("$OUT/bench_$w.json"))
    r = d.get("regime") or {}
    print("$w", "ms/step %.3f" % d["ms_per_step"], "rounds/iter", d["bfgs"]["rounds_per_iter"], "ind_rounds/iter", d["bfgs"]["ind_rounds_per_iter"],
          "kernels", {k: round(v, 3) for k, v in d["per_step_kernel_ms"].items()})
    print("   first iters", [round(x, 1) for x in d["first_iterations_ms"]], d["first_iterations_rounds"])
    print("   versions", {k: round(v["share"], 3) for k, v in (r.get("objective_kernel_versions") or {}).items()})
    print("   est_maf off route", {k: round(v["share"], 5) for k, v in (r.get("est_maf_sites_off_the_common_route") or {}).items()})
except Exception as e:
    print("$w: no line:", e)
PY
done
