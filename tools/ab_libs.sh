#!/bin/bash
# Two or more builds of the library on one box, alternating (NGHMM_LIB): configs[1], one rank of eight,
# the default workload; ms per EM iteration in the steady state.
#   bash tools/ab_libs.sh libnghmm_prev.so libnghmm.so > gpurun_out/ab_libs.log
set -e -o pipefail
line() {
  python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['ms_per_step'],4))"
}
C="--no_cpu_baseline --no_exact_line --no_check"
for rep in 1 2 3; do
  for lib in "$@"; do
    export NGHMM_LIB=$PWD/ngsf-hmm_amd/$lib
    a=$(python3 bench.py --workload c2 --steps 200 --warmup 20 $C | line)
    b=$(python3 bench.py --emulate_ranks 8 --steps 20 --warmup 6 $C | line)
    c=$(python3 bench.py --steps 10 --warmup 6 $C | line)
    echo "rep $rep $lib: c2 $a  rank-of-8 $b  n1 $c"
  done
done
