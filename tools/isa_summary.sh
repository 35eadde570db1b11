#!/bin/bash
# Device assembly of the fast-mode kernels -> profiles/<tag>_isa_summary.txt (instruction counts
# per loop body, registers, occupancy; bench.py parses it).  No GPU needed.
#   bash tools/isa_summary.sh r05
set -e
TAG=${1:-r05}
T=$(mktemp -d)
for f in walks estep estmaf; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only \
      ngsf-hmm_amd/csrc/kernels_fast_$f.hip -o $T/kf_$f.s
done
cat $T/kf_walks.s $T/kf_estep.s $T/kf_estmaf.s > $T/kf.s
python3 tools/isa_report.py $T/kf.s > profiles/${TAG}_isa_summary.txt
rm -rf $T
head -3 profiles/${TAG}_isa_summary.txt
