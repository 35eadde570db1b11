#!/bin/bash
# SQ counters of the planning kernel (k_bfgs_advance) per dispatch on configs[1] (bench.py --workload c2):
# instructions and quad-cycles per wave, what DESIGN.md section 9.2 quotes for "a lone wave on its SIMD".
#   bash profiles/collect_planning_sq.sh r06      (GPU box; writes profiles/<tag>_c2_planning_kernel_sq.json)
set -e -o pipefail
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/planning_sq
rm -rf $OUT
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SMEM \
  SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT -o a -- python3 bench.py --workload c2 \
  --steps 10 --warmup 6 --no_cpu_baseline --no_exact_line --no_check --no_cold --serial_kernels \
  > /dev/null 2> gpurun_out/planning_sq.err
python3 - $TAG <<'PY'
import csv, glob, collections, json, sys, statistics
tag = sys.argv[1]
f = glob.glob("gpurun_out/planning_sq/**/*counter_collection.csv", recursive=True)[0]
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "k_bfgs_advance<false>" not in r["Kernel_Name"].replace("(bool)0", "false"):
        if "k_bfgs_advance" not in r["Kernel_Name"] or "<true>" in r["Kernel_Name"]:
            continue
    d.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
rows = [v for v in d.values() if v.get("SQ_WAVES")]
rows = rows[len(rows) // 2:]   # the steady state: the later iterations of the run
def per_wave(v, k):
    return v.get(k, 0.0) / v["SQ_WAVES"]
keys = {"valu": "SQ_INSTS_VALU", "salu": "SQ_INSTS_SALU", "lds": "SQ_INSTS_LDS", "smem": "SQ_INSTS_SMEM",
        "wave_quad_cycles": "SQ_WAVE_CYCLES", "wait_inst_quad_cycles": "SQ_WAIT_INST_ANY",
        "valu_active_quad_cycles": "SQ_ACTIVE_INST_VALU"}
sys.path.insert(0, "profiles")
from build_id import build_id
out = {"build_id": build_id(), "workload": "c2: 100 individuals x 100 000 sites, one wave per individual, 4 per workgroup",
       "dispatches": len(rows), "unit": "per wave of a dispatch; SQ_WAVE_CYCLES etc. count quad-cycles (4 shader cycles)",
       "median_per_wave": {k: statistics.median(per_wave(v, c) for v in rows) for k, c in keys.items()},
       "max_per_wave": {k: max(per_wave(v, c) for v in rows) for k, c in keys.items()}}
m = out["median_per_wave"]
out["median_cycles_per_instruction"] = 4 * m["wave_quad_cycles"] / (m["valu"] + m["salu"] + m["lds"] + m["smem"])
json.dump(out, open(f"profiles/{tag}_c2_planning_kernel_sq.json", "w"), indent=1)
json.dump(out, open("gpurun_out/c2_planning_kernel_sq.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $OUT
