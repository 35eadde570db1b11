#!/usr/bin/env python3
"""One steady-state EM iteration out of a rocprofv3 kernel trace: every dispatch with its queue,
its start relative to the iteration's first kernel, its duration and the gap to the previous
dispatch's end on the same queue.  An iteration starts at a fresh walk (the emitting k_fast_lkl_fd); the one printed is the 13th of the
trace -- inside the timed loop of `bench.py --steps 10 --warmup 6` (tools/c2_timeline.sh); the last ten of
that run are bench.py's own measuring loops (events on, then one stream), where kernels run one after the other.
  python tools/c2_timeline.py kernel_trace.csv [iteration]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "nghmm" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"nghmm::(\(anonymous namespace\)::)?", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n[:70]
# the fresh walk = the emitting instance of the objective kernel (4th template argument)
starts = [i for i, r in enumerate(rows) if re.search(r"k_fast_lkl_fd<\d+, \d+, \w+, true", r["Kernel_Name"])]
if len(starts) < 4:
    # fall back: the kernel name of the first walk is whatever starts most iterations
    print("no fresh-walk kernels found; kernel names:", sorted({short(r["Kernel_Name"]) for r in rows}))
    sys.exit(0)
which = int(sys.argv[2]) if len(sys.argv) > 2 else 12
a, b = starts[which], starts[which + 1]
t0 = int(rows[a]["Start_Timestamp"])
last_end = {}
print(f"iteration of {b - a} dispatches, {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us from walk to walk")
print(f"{'start us':>9} {'dur us':>8} {'gap us':>8}  q  kernel (grid x wg)")
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
    last_end[q] = e
    busy += e - s
    g = r.get("Grid_Size", r.get("Grid_Size_X", "?")); w = r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:8.1f}  {q}  {short(r['Kernel_Name'])} ({g} x {w})")
print(f"sum of kernel durations {busy / 1e3:.1f} us")
# union of busy intervals
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows[a:b])
u = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: u += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
u += ce - cs
print(f"device busy (union) {u / 1e3:.1f} us")
