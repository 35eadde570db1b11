#!/bin/bash
# One cohort shape on two builds of the library, same box (est_maf variants by cohort size):
#   cp ngsf-hmm_amd/libnghmm.so ngsf-hmm_amd/libnghmm_base.so   # before the change under test
#   bash tools/ab_cohort.sh <individuals> <sites>
for lib in libnghmm_base.so libnghmm.so; do for rep in 1 2; do
NGHMM_LIB=$PWD/ngsf-hmm_amd/$lib python3 bench.py --workload c5k --n_ind $1 --n_sites $2 --steps 20 --warmup 5 --no_cpu_baseline --no_exact_line --no_check --no_cold 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 x $2 $lib', '%.4f ms/step' % d['ms_per_step'], 'est_maf %.3f' % d['per_step_kernel_ms']['est_maf'])"
done; done
