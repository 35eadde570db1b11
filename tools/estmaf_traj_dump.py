#!/usr/bin/env python3
"""Inputs of est_maf for a sample of sites of the benchmark's data set -- linear likelihoods and the
GPU's own posteriors after 1, 2 and 6 EM iterations -- as gpurun_out/estmaf_sample.npz, so that
variants of the interval rule (k_fast_estmaf's build) can be tried offline on the recursion's numpy
restatement (tools/estmaf_offline.py).
  python tools/estmaf_traj_dump.py [n_sites=60000] [every=100] [workload=c3: a bench.py workload whose regime the data take]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("ngsf-hmm_amd")
import torch
S = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 100
I = 1000
dev = torch.device("cuda", 0)
import bench
wl = sys.argv[3] if len(sys.argv) > 3 else "c3"
sim = pkg.simulate.IndexedSim(1000, 1_000_000, dev, seed=12345, **bench.WORKLOADS[wl].get("sim", {}))
gl_d, pos_d = sim.gl((0, I), (0, S)), sim.pos_dist(0, S)
torch.cuda.synchronize()
sites = np.arange(0, S, every)
posts, freqs = {}, {}
with pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST) as fa:
    fa.load_device(gl_d.data_ptr(), pos_d.data_ptr())
    fa.set_params(0.1, 0.2, 0.1)
    fa.init_emission()
    done = 0
    for its in (1, 2, 6):
        while done < its:
            fa.iter_EM()
            done += 1
        fa.estep()
        posts[its] = fa.marg_prob[:, sites].T.copy()     # [n][I]
        fa.mstep_freq(1)
        freqs[its] = fa.freq[sites].copy()
gl = gl_d.cpu().numpy()[sites]                            # [n][I][3] log
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed(f"gpurun_out/estmaf_sample{'' if wl == 'c3' else '_' + wl}.npz", sites=sites, gl=gl.astype(np.float32),
                    post1=posts[1], post2=posts[2], post6=posts[6],
                    freq1=freqs[1], freq2=freqs[2], freq6=freqs[6])
print("saved", len(sites), "sites of", wl)
