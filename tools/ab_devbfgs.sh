#!/bin/bash
# A/B on one box: the M-step's L-BFGS-B machines on the device (default) against the host
# machines (NGHMM_NO_DEV_BFGS=1), configs[1] and the default workload.  Run from the repo root
# on the GPU box:  bash tools/ab_devbfgs.sh   (results: gpurun_out/ab_devbfgs_*.json)
set -e
mkdir -p gpurun_out
B="--steps 10 --warmup 6 --no_cpu_baseline --no_exact_line"
NGHMM_TIMING=1 python bench.py --workload c2 $B > gpurun_out/ab_devbfgs_c2_dev.json 2> gpurun_out/ab_devbfgs_c2_dev.log
NGHMM_NO_DEV_BFGS=1 python bench.py --workload c2 $B > gpurun_out/ab_devbfgs_c2_host.json 2>/dev/null
python bench.py $B > gpurun_out/ab_devbfgs_c3_dev.json 2> gpurun_out/ab_devbfgs_c3_dev.log
NGHMM_NO_DEV_BFGS=1 python bench.py $B > gpurun_out/ab_devbfgs_c3_host.json 2>/dev/null
tail -3 gpurun_out/ab_devbfgs_c2_dev.log
for f in c2_dev c2_host c3_dev c3_host; do
  python - "$f" <<'PY'
import json, sys
f = sys.argv[1]
d = json.loads(open(f"gpurun_out/ab_devbfgs_{f}.json").read().strip().splitlines()[-1])
print(f, round(d["ms_per_step"], 4), "ms/iteration; kernels", d.get("per_step_kernel_ms"), "rounds", d.get("first_iterations_rounds"))
PY
done
