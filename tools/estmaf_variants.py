#!/usr/bin/env python3
"""est_maf build variants (nodes per interval, interval length) on one box: time per EM iteration
of the default workload (sequential kernels) and the frequencies against all-exact passes.
  python tools/estmaf_variants.py libnghmm.so libnghmm_e10_45.so ...      (a child process per library)"""
import importlib, json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    pkg = importlib.import_module("ngsf-hmm_amd")
    import torch
    dev = torch.device("cuda", 0)
    I, S = 1000, 100_000
    sim = pkg.simulate.IndexedSim(1000, 1_000_000, dev, seed=12345)
    gl_d, pos_d = sim.gl((0, I), (0, S)), sim.pos_dist(0, S)
    torch.cuda.synchronize()
    out = {}
    with pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST) as fa:
        fa.load_device(gl_d.data_ptr(), pos_d.data_ptr())
        fa.set_params(0.1, 0.2, 0.1)
        fa.init_emission()
        errs = []
        for it in range(5):
            fa.estep()
            fa.set_switch("estmaf_interp", 1)
            fa.mstep_freq(1)
            f1 = fa.freq.copy()
            # the same step with every pass exact (the frequencies the posteriors belong to are
            # the input's: set them back first)
            fa.set_switch("estmaf_interp", 0)
            fa.mstep_freq(1)
            f0 = fa.freq.copy()
            errs.append(float(np.max(np.abs(f1 - f0) / f0)))
            fa.set_switch("estmaf_interp", 1)
            fa.iter_EM()
        out["freq_vs_exact_max_rel"] = errs
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
        sys.exit(0)
    for lib in sys.argv[1:]:
        env = dict(os.environ, NGHMM_LIB=os.path.join(ROOT, "ngsf-hmm_amd", lib))
        r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        acc = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        acc = json.loads(acc[-1][7:]) if acc else {"error": r.stderr[-400:]}
        b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "6",
                            "--no_cpu_baseline", "--no_exact_line", "--no_check", "--serial_kernels"],
                           env=env, capture_output=True, text=True)
        try:
            d = json.loads([l for l in b.stdout.splitlines() if l.startswith("{")][-1])
            t = {"ms_per_step": round(d["ms_per_step"], 3), "est_maf_ms": round(d["per_step_kernel_ms"]["est_maf"], 3)}
        except Exception:
            t = {"error": b.stderr[-400:]}
        print(lib, json.dumps(t), json.dumps(acc), flush=True)
