#!/usr/bin/env python3
"""Soak test of the asynchronous parts of a fused fast-mode iteration (second stream, epilogue word,
the next M-step's round planned in advance): many iterations with everything on, against the same
iterations with the kernels on one stream, timing events on and a stream synchronisation at the end
(switches no_bg_stream + spans) -- every iteration's arrays bit for bit; then R replicas driven
concurrently from R threads against the same runs one after the other.
   python tools/stress_async.py [iterations=300] [replicas=6]      (needs an MI355X)"""
import importlib, os, sys, threading, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ngsf-hmm_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
R = int(sys.argv[2]) if len(sys.argv) > 2 else 6


def digest(h):
    return tuple(zlib.crc32(np.ascontiguousarray(a).tobytes()) for a in (h.ind_lkl, h.indF, h.alpha, h.freq))


def run(h, start, n, out):
    h.set_params(*start)
    h.init_emission()
    for _ in range(n):
        h.iter_EM()
        out.append(digest(h))


for I, S in ((100, 20_000), (13, 3_000), (700, 6_000)):
    d = pkg.simulate.simulate(I, S, seed=I + S, n_chrom=2, missing_rate=0.03, indF="r", freq="r", alpha=0.2)
    gl = pkg.simulate.normalise_log_gl(d.gl)
    with pkg.NgsFHMM(I, S, mode=pkg.MODE_FAST) as h:
        h.load(gl, d.pos_dist_mb)
        a, b = [], []
        run(h, (0.1, 0.2, 0.1), N, a)
        h.set_switch("no_bg_stream", 1)
        h.set_switch("spans", 1)
        run(h, (0.1, 0.2, 0.1), N, b)
        h.set_switch("no_bg_stream", 0)
        h.set_switch("spans", 0)
        bad = [k for k in range(N) if a[k] != b[k]]
        print(f"{I} x {S}: {N} iterations, asynchronous path vs one stream + events: "
              f"{'identical' if not bad else 'DIFFER from iteration %d' % bad[0]}", flush=True)
        assert not bad
        # replicas: concurrently against one after the other
        rng = np.random.default_rng(1)
        starts = [(rng.uniform(0.05, 0.5, I), rng.uniform(0.05, 0.5, I), rng.uniform(0.05, 0.4, S)) for _ in range(R)]
        reps = [h.replica() for _ in range(R)]
        serial = [[] for _ in range(R)]
        for r in range(R):
            run(reps[r], starts[r], N // 3, serial[r])
        conc = [[] for _ in range(R)]
        th = [threading.Thread(target=run, args=(reps[r], starts[r], N // 3, conc[r])) for r in range(R)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for rp in reps:
            rp.close()
        bad = [(r, k) for r in range(R) for k in range(N // 3) if serial[r][k] != conc[r][k]]
        print(f"{I} x {S}: {R} replicas x {N // 3} iterations, concurrent vs serial: "
              f"{'identical' if not bad else 'DIFFER at %s' % (bad[0],)}", flush=True)
        assert not bad
print("stress ok")
