#!/bin/bash
# The same bench lines on several builds of the library (NGHMM_LIB), same box:
#   bash tools/ab_libs.sh "<lib> <lib> ..." "<workload> ..." [steps]
LIBS=$1; WLS=$2; STEPS=${3:-20}
OUT=gpurun_out/ab_libs.log
: > $OUT
for w in $WLS; do
  for lib in $LIBS; do
    for rep in 1 2; do
      NGHMM_LIB=$PWD/ngsf-hmm_amd/$lib python3 bench.py --workload $w --steps $STEPS --warmup 5 --no_cpu_baseline --no_exact_line --no_check --no_cold \
        2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=(d.get('regime') or {}).get('est_maf_sites_off_the_common_route') or {}; print('$w $lib', '%.4f ms/step' % d['ms_per_step'], 'est_maf %.3f' % d['per_step_kernel_ms']['est_maf'], {k: round(v['share'],4) for k,v in r.items() if v['share']})" | tee -a $OUT
    done
  done
done
