#!/bin/bash
# A/B of one library switch on one box (here: spans, the timing events of a fused iteration; earlier:
# stream priorities, a CU-masked second stream -- no gain): configs[1], one rank of eight, the default workload, alternating.
#   bash tools/ab_prio.sh > gpurun_out/ab_prio.log
set -e -o pipefail
line() {  # ms_per_step of a bench run
  python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['ms_per_step'],4))"
}
C="--no_cpu_baseline --no_exact_line --no_check"
for rep in 1 2 3; do
  for p in 0 1; do
    export NGHMM_NO_BG_STREAM=$p
    a=$(python3 bench.py --workload c2 --steps 200 --warmup 20 $C | line)
    b=$(python3 bench.py --emulate_ranks 8 --steps 20 --warmup 6 $C | line)
    c=$(python3 bench.py --steps 10 --warmup 6 $C | line)
    echo "rep $rep no_bg_stream $p: c2 $a  rank-of-8 $b  n1 $c"
  done
done
