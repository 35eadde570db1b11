#!/bin/bash
# A/B of round 6's changes to the iteration's critical path on small cohorts (same box, same build):
#   bash tools/ab_round6.sh [workload] [steps] [VAR=1 ...]   (each VAR=1 is one more variant)
WL=${1:-c2}; STEPS=${2:-300}; shift; shift
OUT=gpurun_out/ab_round6_$WL.log
: > $OUT
run() {
  local name=$1; shift
  for rep in 1 2; do
    env "$@" python3 bench.py --workload $WL --steps $STEPS --warmup 20 --no_cpu_baseline --no_exact_line --no_check \
      2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$WL $name', '%.4f ms/step' % d['ms_per_step'], 'rounds/iter', d['bfgs']['rounds_per_iter'])" | tee -a $OUT
  done
}
run default NGHMM_X=0
for v in "$@"; do run "$v" "$v"; done
