# bench the default workload with variant builds of the library (tools: kernel tuning)
#   usage: bash tools/sweep_libs.sh "" _en12 _deg2
for v in "$@"; do
  export NGHMM_LIB=$PWD/ngsf-hmm_amd/libnghmm$v.so
  echo "== $NGHMM_LIB"
  python bench.py --steps 8 --warmup 4 --no_cpu_baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['per_step_kernel_ms'].items()}, {k: round(v,2) for k,v in d['bfgs'].items()})"
done
