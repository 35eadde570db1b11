#!/usr/bin/env python3
"""A/B of the objective kernels' operator tree: the LDS-packed tree over all points of a group
(default build) against a shuffle tree per point (ngsf-hmm_amd/libnghmm_shfl.so, built with
-DNGHMM_TREE_SHFL): the same bits, and the round's kernel time at several shapes.
   make -C ngsf-hmm_amd/csrc ../libnghmm_shfl.so && python tools/ab_tree.py     (needs an MI355X)"""
import importlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import importlib, os, sys, json
import numpy as np
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("ngsf-hmm_amd")
out = {}
for I, S in ((1000, 125_000), (1000, 1_000_000), (100, 100_000)):
    sim = pkg.simulate.IndexedSim(I, S, torch.device("cuda", 0), seed=12345)
    gl, pos = sim.gl(), sim.pos_dist(0, S)
    torch.cuda.synchronize()
    with pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST) as h:
        h.load_device(gl.data_ptr(), pos.data_ptr())
        del gl
        h.set_params(0.1, 0.2, 0.1)
        h.init_emission()
        h.estep()
        eh = 4e-6
        ind = np.repeat(np.arange(I), 5).astype(np.uint32)
        F = np.tile([0.1, 0.1 + eh, 0.1 - eh, 0.1, 0.1], I)
        A = np.tile([0.2, 0.2, 0.2, 0.2 + eh, 0.2 - eh], I)
        ms = []
        for k in range(12):
            v = h.lkl(ind, F, A)
            ms.append(h.kernel_ms("lkl_batch")[0])
        for it in range(3):
            h.iter_EM()
        out[f"{I}x{S}"] = dict(round_ms=min(ms), lkl=v[:7].tolist() + [float(v.sum())], indF=float(h.indF.sum()),
                               freq=float(h.freq.sum()), tot=float(h.ind_lkl.sum()))
print(json.dumps(out))
'''
res = {}
for name, lib in (("lds_tree", ""), ("shuffle_tree", os.path.join(ROOT, "ngsf-hmm_amd", "libnghmm_shfl.so"))):
    env = dict(os.environ)
    if lib:
        env["NGHMM_LIB"] = lib
    r = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + CHILD], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    res[name] = json.loads(r.stdout.strip().splitlines()[-1])
for shape in res["lds_tree"]:
    a, b = res["lds_tree"][shape], res["shuffle_tree"][shape]
    same = a["lkl"] == b["lkl"] and a["indF"] == b["indF"] and a["freq"] == b["freq"] and a["tot"] == b["tot"]
    print(f"{shape}: round of 5 points {b['round_ms']:.3f} ms (shuffle tree per point) -> {a['round_ms']:.3f} ms "
          f"(packed LDS tree); values, and three EM iterations after them, bit-identical: {same}")
