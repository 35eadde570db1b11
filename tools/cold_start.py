#!/usr/bin/env python3
"""The first iterations of a run (the cold start BASELINE's "EM iters/sec" includes): wall time,
objective rounds and kernel-family milliseconds of every iteration from the starting values, and
the kernel versions of its rounds from nghmm_debug_mode_counts.
   python tools/cold_start.py [workload [iterations]]     (needs an MI355X)"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
pkg = importlib.import_module("ngsf-hmm_amd")
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 8
I, S = wl["n_ind"], wl["n_sites"]
dev = torch.device("cuda", 0)
mode = pkg.MODE_FAST | (pkg.GENO_PACKED if wl.get("call_geno") else 0)
sim = pkg.simulate.IndexedSim(I, S, dev, seed=12345, n_chrom=wl.get("n_chrom", 1))
with pkg.NgsFHMM(I, S, mode=mode) as h:
    h.set_switch("spans", 1)    # kernel times of the fused iterations (off by default)
    pos = sim.pos_dist(0, S)
    if wl.get("call_geno"):
        def feed():
            for a, c in sim.chunks(chunk_sites=50_000):
                torch.cuda.synchronize()
                yield a, c.shape[0], c.data_ptr()
        h.load_chunks_device(pos.data_ptr(), feed(), space=0, call_geno=True)
    else:
        gl = sim.gl()
        torch.cuda.synchronize()
        h.load_device(gl.data_ptr(), pos.data_ptr())
        del gl
    for rep in range(2):            # the second run starts with everything allocated and touched
        h.set_params(0.1, 0.2, 0.1)
        h.init_emission()
        rows = []
        for it in range(n_it):
            t = time.perf_counter()
            st = h.iter_EM()
            ms = (time.perf_counter() - t) * 1e3
            fam = {k: round(h.kernel_ms(k)[0], 2) for k in ("lkl_first", "lkl_batch", "forward", "est_maf")}
            rows.append(dict(it=it + 1, ms=round(ms, 2), rounds=int(st.rounds), ind_rounds=int(st.ind_rounds),
                             points=int(st.points), **fam))
            sys.stderr.write(f"--- run {rep} iteration {it + 1} done\n")
        print(json.dumps({"run": rep, "iterations": rows, "mean_ms": sum(r["ms"] for r in rows) / n_it}))
