#!/bin/bash
# A/B on one box: the objective kernels' SMALL versions in the kappa form (libnghmm.so) against
# the c form (libnghmm_kf0.so: make -C ngsf-hmm_amd/csrc ../libnghmm_kf0.so), alternating.
set -e
mkdir -p gpurun_out
B="--steps 10 --warmup 6 --no_cpu_baseline --no_exact_line --no_check"
run() { # name lib args...
  name=$1; lib=$2; shift; shift
  NGHMM_LIB=$PWD/ngsf-hmm_amd/$lib python bench.py $B "$@" > gpurun_out/abk_$name.json 2> gpurun_out/abk_$name.err
  python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
d = json.loads(open(f"gpurun_out/abk_{n}.json").read().strip().splitlines()[-1])
k = d.get("per_step_kernel_ms") or {}
print(f"{n:28s} {d['ms_per_step']:8.4f} ms/iteration  kernels(seq) " + " ".join(f"{a}={b:.3f}" for a, b in k.items()), flush=True)
PY
}
for rep in 1 2; do
  run c3_kf0_$rep libnghmm_kf0.so
  run c3_kf1_$rep libnghmm.so
done
run c3_serial_kf0 libnghmm_kf0.so --serial_kernels
run c3_serial_kf1 libnghmm.so --serial_kernels
run c2_kf0 libnghmm_kf0.so --workload c2 --steps 100 --warmup 10
run c2_kf1 libnghmm.so --workload c2 --steps 100 --warmup 10
run r8_kf0 libnghmm_kf0.so --emulate_ranks 8
run r8_kf1 libnghmm.so --emulate_ranks 8
