#!/bin/bash
# SQ counters of est_maf on a likelihood cohort above 4096 individuals (bench.py --workload c5k),
# next to the committed ones of the default workload: what DESIGN.md section 9.1 quotes for "a CU holds
# one site".   bash profiles/collect_cohort_sq.sh r06      (GPU box; writes profiles/<tag>_c5k_estmaf_sq.json)
set -e -o pipefail
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/c5k_sq
rm -rf $OUT
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU \
  GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT -o sq -- python3 bench.py \
  --workload c5k --steps 2 --warmup 1 --no_cpu_baseline --no_exact_line --no_check --no_cold --serial_kernels \
  > /dev/null 2> gpurun_out/c5k_sq.err
python3 - $TAG <<'PY'
import csv, glob, collections, json, re, sys
tag = sys.argv[1]
f = glob.glob("gpurun_out/c5k_sq/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "k_fast_estmaf<" not in k:
        continue
    k = re.search(r"k_fast_estmaf<[^>]*>", k).group(0)
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        n[k] += 1
sys.path.insert(0, "profiles")
from build_id import build_id
out = {"build_id": build_id(), "workload": "c5k: 5000 individuals x 100 000 sites, est_maf in two parts per iteration",
       "cells_per_launch": 5000 * 50000}
for k, c in acc.items():
    wc = c["SQ_WAVE_CYCLES"]
    out[k] = {"launches": n[k], "valu_active_frac_per_wave": c["SQ_ACTIVE_INST_VALU"] / wc,
              "wait_inst_frac": c["SQ_WAIT_INST_ANY"] / wc, "wait_any_frac": c["SQ_WAIT_ANY"] / wc,
              "insts_valu_per_launch": c["SQ_INSTS_VALU"] / n[k],
              "insts_valu_per_cell": c["SQ_INSTS_VALU"] / n[k] / (5000 * 50000)}
ref = json.load(open(sorted(glob.glob("profiles/r[0-9][0-9]_pmc_summary.json"))[-1]))
k = ref.get("k_fast_estmaf<16, 64, true>", {})
out["default_workload_k_fast_estmaf<16, 64, true>"] = {
    a: k.get(a) for a in ("valu_active_frac_per_wave", "wait_inst_frac", "wait_any_frac", "insts_valu_per_launch")}
if k.get("insts_valu_per_launch"):
    out["default_workload_k_fast_estmaf<16, 64, true>"]["insts_valu_per_cell"] = k["insts_valu_per_launch"] / (1000 * 500000)
json.dump(out, open(f"profiles/{tag}_c5k_estmaf_sq.json", "w"), indent=1)
json.dump(out, open("gpurun_out/c5k_estmaf_sq.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $OUT
