# same-box A/B of round 3's fast-mode changes (ms per EM iteration at 1000 x 1M):
#   old: 14 est_maf nodes, degree-4 alpha probes (build ngsf-hmm_amd/libnghmm_en14.so with
#        -DNGHMM_EST_EN=14 -DNGHMM_EST_TOL=1e-13 first)
#   new: the defaults
run() { python bench.py --steps 10 --warmup 4 --no_cpu_baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['per_step_kernel_ms'].items()})"; }
for rep in 1 2; do
  NGHMM_LIB=$PWD/ngsf-hmm_amd/libnghmm_en14.so NGHMM_NO_XDEG2=1 run "old           "
  NGHMM_NO_XDEG2=1 run "new: en12     "
  run "new: +xdeg2   "
done
