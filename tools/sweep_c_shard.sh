#!/bin/bash
# waves per individual (NGHMM_FAST_C) at the per-rank shapes of a site-sharded 1000 x 1M run
for S in 124992 250000 500000; do
  for C in 0 8 12 16 24 32 40; do
    if [ $C = 0 ]; then unset NGHMM_FAST_C; else export NGHMM_FAST_C=$C; fi
    python bench.py --no_cpu_baseline --n_sites $S --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('S=$S C=$C', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['per_step_kernel_ms'].items()}, d['bfgs']['rounds_per_iter'])
" || exit 1
  done
done
