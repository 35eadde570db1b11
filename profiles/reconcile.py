#!/usr/bin/env python3
"""Set rocprofv3's per-kernel durations against bench.py's per-family HIP-event timings of
the SAME run (profiles/collect.sh runs bench.py under `rocprofv3 --kernel-trace --stats`
with --warmup 0, so the trace holds exactly the timed kernels).

bench.py times kernel FAMILIES (nghmm_kernel_ms): `lkl_batch` = one objective round = the
k_fast_lkl_fd<...> / k_fast_lkl_chunks<...> launches of that round + k_fast_lkl_finish;
`forward` = k_fast_bounds + k_fast_bwd_recompute8 (+ k_fast_chunk_ops outside
nghmm_estep_mstep); `est_maf` = k_fast_estmaf<...> (+ _resume x 2) + k_fast_estmaf_interp x 2 +
k_fast_estmaf_stream + k_fast_freq_interleave's neighbour kernels.  The family's average per
launch is therefore  sum(TotalDurationNs of its kernels) / (rounds or EM iterations).

usage: reconcile.py <kernel_stats.csv> <bench_under_trace.json>
"""
import csv
import json
import re
import sys

FAMILIES = {
    "lkl_batch": ("k_fast_lkl_fd", "k_fast_lkl_chunks", "k_fast_lkl_finish"),
    "forward": ("k_fast_chunk_ops", "k_fast_bounds", "k_fast_bwd_recompute",
                "k_fast_bwd_recompute8"),
    "est_maf": ("k_fast_estmaf", "k_fast_estmaf_resume", "k_fast_estmaf_rows",
                "k_fast_estmaf_rows_resume", "k_fast_estmaf_interp", "k_fast_estmaf_stream",
                "k_fast_post_to_site_major"),
    "bfgs": ("k_bfgs_advance",),
}


def short(name):
    m = re.search(r"(k_\w+)(?:<[^>(]*>)?\(", name)
    return m.group(1) if m else name


def main():
    stats, bench = sys.argv[1:3]
    tot, calls = {}, {}
    for r in csv.DictReader(open(stats)):
        if "nghmm" not in r["Name"]:
            continue
        k = short(r["Name"])
        tot[k] = tot.get(k, 0.0) + float(r["TotalDurationNs"]) / 1e6
        calls[k] = calls.get(k, 0) + int(r["Calls"])
    b = json.load(open(bench))
    iters = calls.get("k_fast_bwd_recompute8", 0) or calls.get("k_fast_bwd_recompute", 0)
    # (round 6: the planning kernel finishes its own individual's points -- no k_fast_lkl_finish in
    # device-planned rounds; one k_bfgs_advance per round + one per M-step that plans round 1)
    rounds = calls.get("k_fast_lkl_finish", 0) or max(calls.get("k_bfgs_advance", 0) - iters, 0)
    out = {"em_iterations_in_trace": iters, "objective_rounds_in_trace": rounds,
           "bench_steps": b["steps"], "families": {}}
    for fam, kernels in FAMILIES.items():
        ms = sum(tot.get(k, 0.0) for k in kernels)
        n = rounds if fam == "lkl_batch" else iters
        out["families"][fam] = {
            "rocprof_total_ms": ms,
            "rocprof_ms_per_launch": ms / n if n else None,
            "rocprof_ms_per_em_iteration": ms / iters if iters else None,
            "bench_ms_per_em_iteration": b["per_step_kernel_ms"].get(fam),
            "bench_avg_launch_ms": b["roofline_all_kernels"].get(fam, {}).get("avg_launch_ms"),
        }
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
