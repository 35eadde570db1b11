#!/bin/bash
# Round-5 A/B on one box (results: gpurun_out/ab5_*.json):
#   default / c2 / one rank of 8 (site shards): L-BFGS-B machines on the device vs on the host
#   est_maf on two waves per site (switch estmaf_w2) vs one, sequential kernels
set -e
mkdir -p gpurun_out
B="--steps 10 --warmup 6 --no_cpu_baseline --no_exact_line --no_check"
run() { # name env... -- args
  name=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" python bench.py $B "$@" > gpurun_out/ab5_$name.json 2> gpurun_out/ab5_$name.err
  python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
d = json.loads(open(f"gpurun_out/ab5_{n}.json").read().strip().splitlines()[-1])
k = d.get("per_step_kernel_ms") or {}
print(f"{n:28s} {d['ms_per_step']:8.4f} ms/iteration  kernels(seq) " + " ".join(f"{a}={b:.3f}" for a, b in k.items()))
PY
}
run c3_dev X=1 --
run c3_host NGHMM_NO_DEV_BFGS=1 --
run c3_dev_onestream NGHMM_NO_BG_STREAM=1 --
run c2_dev X=1 -- --workload c2
run c2_host NGHMM_NO_DEV_BFGS=1 -- --workload c2
run r8_dev X=1 -- --emulate_ranks 8
run r8_host NGHMM_NO_DEV_BFGS=1 -- --emulate_ranks 8
run c3_serial X=1 -- --serial_kernels
run c3_serial_estmaf_w2 NGHMM_ESTMAF_W2=1 -- --serial_kernels
