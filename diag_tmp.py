import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module("ngsf-hmm_amd")
for I, S, kw in ((1000, 200000, {}), (1000, 200000, dict(freq="r")), (1000, 100000, dict(freq="r", depth=10.0)), (2000, 50000, dict(freq="r")), (100, 200000, dict(freq="r"))):
    gl, pos = pkg.simulate.simulate_torch(I, S, "cuda", seed=1, **kw)
    hmm = pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST)
    hmm.load_device(gl.data_ptr(), pos.data_ptr())
    res = {}
    for interp in ("0", "1"):
        os.environ["NGHMM_ESTMAF_INTERP"] = interp
        hmm.set_params(np.full(I, 0.1), np.full(I, 0.01), np.full(S, 0.1))
        hmm.init_emission(); hmm.estep()
        for it in range(2):
            hmm.mstep_freq(1)
        ms, n = hmm.kernel_ms("est_maf")
        res[interp] = (hmm.freq.copy(), ms)
    f0, f1 = res["0"][0], res["1"][0]
    rel = np.abs(f1 - f0) / np.abs(f0)
    print(I, S, kw, "exact ms", round(res["0"][1], 3), "interp ms", round(res["1"][1], 3), "max rel diff", rel.max(), "n>1e-12", int((rel > 1e-12).sum()), flush=True)
    hmm.close(); del gl, pos
