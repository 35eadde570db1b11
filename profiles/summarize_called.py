#!/usr/bin/env python3
"""The called genotypes' est_maf kernels on config 5's rank (5000 individuals x 625 000 sites,
packed): launch duration from the kernel trace, HBM bytes from separate FETCH_SIZE / WRITE_SIZE
passes (KiB; FETCH_SIZE raw and doubled as MI355X_MICROARCH.md prescribes for wide streams --
this kernel's loads are 8 B and 4 B per lane, so the raw figure is the closer one), against the
algorithmic 8.25 B per cell.
usage: summarize_called.py <kernel_stats.csv> <fetch.csv> <write.csv> <n_ind> <n_sites> [iterations]
(the frequency step goes behind the objective rounds in parts: a launch covers 1 / parts of the
tile rows; parts = calls / iterations of the traced run)"""
import csv
import json
import sys

stats, fetch, write, I, S = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 3
out = {}
for r in csv.DictReader(open(stats)):
    name = r.get("Name") or r.get("Kernel_Name") or ""
    if "k_fast_estmaf_called" in name:
        key = "sums" if "called_sums" in name else "passes"
        out[key] = {"kernel": name.split("(")[0][-60:], "calls": int(r["Calls"]),
                    "avg_ms": float(r["AverageNs"]) / 1e6}


def pmc(path, counter):
    tot, ids = 0.0, set()
    for r in csv.DictReader(open(path)):
        if "k_fast_estmaf_called_sums" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
            ids.add(r["Dispatch_Id"])
    return tot * 1024.0 / max(len(ids), 1), len(ids)


f, nf = pmc(fetch, "FETCH_SIZE")
w, nw = pmc(write, "WRITE_SIZE")
parts = max(1, round(out["sums"]["calls"] / iters))
algo = (8.0 + 0.25) * I * S / parts
ms = out.get("sums", {}).get("avg_ms")
out["sums"].update({
    "launches_per_em_iteration": parts, "ms_per_em_iteration": (ms or 0) * parts,
    "launches_in_pmc_passes": [nf, nw], "fetch_bytes_raw_per_launch": f,
    "fetch_bytes_doubled_per_launch": 2 * f, "write_bytes_per_launch": w,
    "algorithmic_bytes_per_launch": algo, "traffic_over_algorithmic_raw": (f + w) / algo,
    "traffic_over_algorithmic_doubled_fetch": (2 * f + w) / algo,
    "achieved_GBps_algorithmic": algo / (ms * 1e-3) / 1e9 if ms else None,
    "hbm_frac_of_8TBps": algo / (ms * 1e-3) / 8e12 if ms else None,
    "ps_per_cell": ms * parts * 1e-3 / (I * S) * 1e12 if ms else None})
print(json.dumps(out, indent=1))
