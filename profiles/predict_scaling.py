#!/usr/bin/env python3
"""Predicted strong-scaling points of the 1000 x 1M job from ONE GPU: the emulated-rank lines of
profiles/collect_all.sh (a rank's compute; all-gathers as local copies, or through a one-rank RCCL
group) against the one-GPU line of the same session.  Not a measurement of N GPUs.

  python profiles/predict_scaling.py [tag]   ->  profiles/<tag>_predicted_scaling.json
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"


def line(name):
    with open(os.path.join(HERE, f"{tag}_{name}.json")) as fh:
        return json.loads(fh.read().strip().splitlines()[-1])


one = line("bench_n1_steps20")
out = {"what": "predicted from one GPU: a rank's measured compute per EM iteration of a V-rank strong-scaling "
               "run of 1000 x 1M (bench.py --emulate_ranks V), not a measurement of V GPUs",
       "n1_ms_per_iteration": one["ms_per_step"], "n1_value": one["value"], "points": {}}
for v in (2, 4, 8):
    row = {}
    for kind in ("sites", "sites_rccl_in_loop", "individuals"):
        d = line(f"bench_rank_of_{v}_{kind}")
        ms = d["ms_per_step"]
        row[kind] = {"rank_ms_per_iteration": ms,
                     "job_site_ind_updates_per_s_if_exchange_hidden": one["value"] * one["ms_per_step"] / ms,
                     "efficiency": one["ms_per_step"] / (v * ms)}
    # the individual shards' exchange on point-to-point links
    gb_per_link = 8.0 * (1000 / v) * 1e6 / v / 1e9          # posteriors of one rank for one peer
    row["individuals"]["posterior_GB_per_link_and_iteration"] = gb_per_link
    row["individuals"]["link_ms_at_64_GBps"] = gb_per_link / 64.0 * 1e3
    out["points"][str(v)] = row
path = os.path.join(HERE, f"{tag}_predicted_scaling.json")
with open(path, "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps(out["points"], indent=1))
