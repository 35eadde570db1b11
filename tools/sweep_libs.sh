for v in "" _en8; do
  export NGHMM_LIB=$PWD/ngsf-hmm_amd/libnghmm$v.so
  echo "== $NGHMM_LIB"
  python bench.py --steps 6 --warmup 3 --no_cpu_baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['per_step_kernel_ms'].items()})"
  NGHMM_ESTMAF_INTERP=0 python bench.py --steps 3 --warmup 1 --no_cpu_baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('all exact', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['per_step_kernel_ms'].items()})"
done
