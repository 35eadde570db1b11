#!/bin/bash
# waves per individual (NGHMM_FAST_C) against ms per EM iteration: bash tools/sweep_c.sh "<C values>" [bench args]
CS=${1:-"40 49 56 64"}; shift
for c in $CS; do
  NGHMM_FAST_C=$c python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_exact_line --no_check "$@" > gpurun_out/sweepc_$c.json 2>/dev/null
  python - $c <<'PY'
import json,sys
c=sys.argv[1]
d=json.loads(open(f"gpurun_out/sweepc_{c}.json").read().strip().splitlines()[-1])
print(c, round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['per_step_kernel_ms'].items()}, flush=True)
PY
done
