#!/bin/bash
# Exact mode at 1000 x 1M: how many waves per SIMD est_maf may hold while it runs underneath the
# objective rounds (NGHMM_EXACT_BG_WAVES: 0 = all eight, 4 / 3 / 2, -1 = after the rounds) and how
# many of its 16 pieces may be queued underneath a round (NGHMM_EXACT_BG_DEPTH).
for wd in ${1:-3:2}; do
  w=${wd%%:*}; d=${wd##*:}
  echo "== exact_bg_waves $w exact_bg_depth $d"
  NGHMM_EXACT_BG_WAVES=$w NGHMM_EXACT_BG_DEPTH=$d python tools/exact_timing.py 1000 1000000 4 0 2>&1 | grep -E "^pc"
done
