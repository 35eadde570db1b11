#!/bin/bash
# Copies what profiles/collect_all.sh left under gpurun_out/prof_<tag>/ into profiles/ (the tracked
# summaries) and recomputes the predicted scaling.   bash profiles/install.sh r05
set -e
TAG=${1:-r05}
O=gpurun_out/prof_$TAG
cp $O/check_n1.json profiles/check_n1.json
cp $O/summary.txt profiles/${TAG}_rocprof_summary.txt
mkdir -p profiles/${TAG}_pmc && cp $O/pmc/*.csv profiles/${TAG}_pmc/
for f in $O/${TAG}_*.json $O/${TAG}_*.csv; do cp $f profiles/; done
python3 profiles/predict_scaling.py $TAG > /dev/null
python3 - $TAG <<'EOF'
import glob, json, sys
tag = sys.argv[1]
for f in sorted(glob.glob(f"profiles/{tag}_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print(f"{f.split('/')[-1]:48s} {d['ms_per_step']:9.3f} ms  {d['value']:.3e}  frac {r.get('frac')}  traffic {r.get('traffic')}")
    except Exception as e:
        print(f, "unreadable:", e)
EOF
