"""The multi-rank EM orchestration (ngsf-hmm_amd/distributed.py) on CPU: two processes,
gloo backend, a CPU stand-in for the GPU backend built on the oracle.  Individuals are
sharded over the ranks; posteriors travel by all-to-all into site-sharded blocks,
frequencies by all-gather.  The result must equal a single-process run over all
individuals BIT FOR BIT (est_maf sums individuals in global order: rank-major)."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import importlib, os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orclib
pkg = importlib.import_module("ngsf-hmm_amd")
dd = importlib.import_module("ngsf-hmm_amd.distributed")

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
pf = dd.preflight()                       # known-answer all-to-all / all-gather first
assert pf["world"] == world and pf["backend"] == "gloo"
if os.environ.get("NGHMM_TEST_CORRUPT"):
    # a collective that delivers one wrong element must stop the run
    real = dd.all_to_all if os.environ["NGHMM_TEST_CORRUPT"] == "a2a" else dd.all_gather
    def corrupt(out, inp):
        real(out, inp)
        if rank == 1:
            out.view(-1)[3] += 1e-9
    setattr(dd, "all_to_all" if os.environ["NGHMM_TEST_CORRUPT"] == "a2a" else "all_gather", corrupt)
    try:
        dd.preflight()
    except RuntimeError as e:
        print("PREFLIGHT_CAUGHT rank", rank, str(e)[:120], flush=True)
    else:
        print("PREFLIGHT_PASSED rank", rank, flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0)
I_loc, S = 5, 240
d = pkg.simulate.simulate(I_loc * world, S, seed=31, n_chrom=2, missing_rate=0.05)
PACKED = os.environ.get("NGHMM_TEST_PACKED") == "1"
orc = orclib.Oracle("libm")
gl_all = orc.prepare_gl(d.gl, 0, call_geno=True) if PACKED else pkg.simulate.normalise_log_gl(d.gl)
gl_loc = np.ascontiguousarray(gl_all[:, rank * I_loc:(rank + 1) * I_loc, :])
# the four classes of a called genotype as the preparation leaves them (genotype 0, 1, 2, uniform)
PROTO = orc.prepare_gl(np.array([[0, -1e15, -1e15], [-1e15, 0, -1e15], [-1e15, -1e15, 0],
                                 [-1.0, -1.0, -1.0]]), 0, call_geno=True)


class OracleBackend:
    """Same interface as distributed.GpuBackend, computing with the CPU oracle."""
    def __init__(self):
        self.em = None
    def empty(self, *shape):
        return torch.empty(shape, dtype=torch.float64)
    packed = PACKED
    def load_device(self, gl, pos):
        self.em = orclib.OracleEM(orc, gl.numpy(), pos.numpy())
        self.pos = pos.numpy()
    def load_chunks_device(self, pos, chunks, space=0, call_geno=False):
        raw = np.concatenate([c.numpy() for _, c in sorted(chunks, key=lambda t: t[0])])
        self.gl = orc.prepare_gl(raw, space, call_geno=call_geno)
        self.load_device(torch.from_numpy(self.gl), pos)
    def geno_codes(self):
        codes = np.full(self.gl.shape[:2], 255, dtype=np.uint8)
        for k in range(4):
            codes[(self.gl == PROTO[k]).all(axis=2)] = k
        assert (codes < 4).all()
        return torch.from_numpy(codes)
    def load_geno_site_shard_device(self, codes):
        self.gl_shard = PROTO[codes.numpy()]                     # [S_own][I_tot][3]
    def set_params(self, indF, alpha, freq):
        self.em.set_params(indF, alpha, freq)
    def init_emission(self):
        assert self.em.init_emission() == 0
    def estep(self):
        assert self.em.estep() == 0
        return self.em.ind_lkl
    def mstep_indf(self, a, b):
        assert self.em.mstep_indf(a, b) == 0
        return None
    def shard_config(self, I_tot, ind_begin, site_begin, S_own):
        self.I_tot, self.site_begin, self.S_own = I_tot, site_begin, S_own
    def load_site_shard_device(self, shard):
        self.gl_shard = shard.numpy().copy()                     # [S_own][I_tot][3]
    def pack_posteriors(self, lo, hi, out):
        out.view(hi - lo, -1).copy_(
            torch.from_numpy(np.ascontiguousarray(self.em.marg.T[lo:hi])))         # [sites][I_loc]
    def mstep_freq_sites(self, blocks, freq_out):
        b = blocks.numpy()                                       # [rank][S_own][I_loc]
        for s in range(self.S_own):
            F = np.concatenate([b[q, s] for q in range(b.shape[0])])   # global individual order
            freq_out[s] = orc.est_maf(self.gl_shard[s], F)[0]
    def set_freq(self, freq_all):
        self.em.set_params(None, None, freq_all.numpy())
        assert self.em.init_emission() == 0                      # emission refresh from freq


be = OracleBackend()
em = dd.ShardedEM(pkg, I_loc, S, rank=rank, world=world, backend=be)
if PACKED:   # chunked loading + the one-off exchange of genotype codes
    raw_loc = np.ascontiguousarray(d.gl[:, rank * I_loc:(rank + 1) * I_loc, :])
    cuts = [0, 7, 100, S]
    em.load_chunks_device(torch.from_numpy(d.pos_dist_mb.copy()),
                          [(lo, torch.from_numpy(raw_loc[lo:hi])) for lo, hi in zip(cuts[:-1], cuts[1:])],
                          space=0, call_geno=True)
else:
    em.load_device(torch.from_numpy(gl_loc), torch.from_numpy(d.pos_dist_mb.copy()))
em.set_params(0.1, 0.2, 0.1)
em.init_emission()
for it in range(3):
    em.iter_EM()
    tm = em.timing                        # exchange accounting of bench.py's N > 1 line
    assert tm["iterations"] == it + 1 and tm["freq_step_ms"] > 0 and tm["allgather_ms"] > 0
assert em.collective_bytes_per_iter() == {"all_to_all_out": 8 * (S // world) * I_loc * (world - 1),
                                          "all_gather_out": 8 * (S // world) * (world - 1)}
np.savez(os.path.join(OUT, f"rank{rank}.npz"), indF=be.em.indF, alpha=be.em.alpha,
         freq=be.em.freq, marg=be.em.marg, ind_lkl=be.em.ind_lkl)
dist.destroy_process_group()
'''


@pytest.mark.parametrize("packed", [False, True])
def test_two_rank_sharded_em_equals_single_process(tmp_path, pkg, orc_libm, packed):
    """packed: the called-genotype path -- chunked loading, site shards built from exchanged
    genotype codes (ShardedEM.load_chunks_device / _exchange_code_shard)."""
    import orclib
    world = 2
    script = tmp_path / "worker.py"
    script.write_text(f"ROOT = {ROOT!r}\nOUT = {str(tmp_path)!r}\n" + WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29612" if packed else "29611",
               WORLD_SIZE=str(world), NGHMM_TEST_PACKED="1" if packed else "0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)))
             for r in range(world)]
    try:
        for p in procs:
            assert p.wait(timeout=600) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()

    I_loc, S = 5, 240
    d = pkg.simulate.simulate(I_loc * world, S, seed=31, n_chrom=2, missing_rate=0.05)
    gl = orc_libm.prepare_gl(d.gl, 0, call_geno=True) if packed else pkg.simulate.normalise_log_gl(d.gl)
    ref = orclib.OracleEM(orc_libm, gl, d.pos_dist_mb)
    ref.set_params(0.1, 0.2, 0.1)
    ref.init_emission()
    for _ in range(3):
        assert ref.iterate() == 0
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        sl = slice(r * I_loc, (r + 1) * I_loc)
        assert np.array_equal(got["indF"], ref.indF[sl])
        assert np.array_equal(got["alpha"], ref.alpha[sl])
        assert np.array_equal(got["freq"], ref.freq)
        assert np.array_equal(got["marg"], ref.marg[sl])
        assert np.array_equal(got["ind_lkl"], ref.ind_lkl[sl])


@pytest.mark.parametrize("which", ["a2a", "gather"])
def test_collective_preflight_catches_a_wrong_element(tmp_path, which):
    """distributed.preflight (what bench.py runs before loading any data, also under nccl =
    RCCL): one element off by 1e-9 on one rank, in either collective, raises there."""
    world = 2
    script = tmp_path / "worker.py"
    script.write_text(f"ROOT = {ROOT!r}\nOUT = {str(tmp_path)!r}\n" + WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613" if which == "a2a" else "29614",
               WORLD_SIZE=str(world), NGHMM_TEST_CORRUPT=which)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    try:
        outs = [p.communicate(timeout=120)[0] for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), outs
    assert "PREFLIGHT_CAUGHT rank 1" in outs[1], outs[1]
    assert ("all_to_all_single" if which == "a2a" else "all_gather_into_tensor") in outs[1]
    assert "PREFLIGHT_PASSED rank 0" in outs[0], outs[0]


def test_site_ranges(pkg):
    dd = importlib.import_module("ngsf-hmm_amd.distributed")
    assert dd.site_ranges(12, 4) == [(0, 3), (3, 6), (6, 9), (9, 12)]
    with pytest.raises(ValueError):
        dd.site_ranges(10, 4)


# ---- site shards (fast mode): ngsf-hmm_amd/distributed.py SiteExchange / SiteShardedEM ----
SITE_WORKER = r'''
import importlib, os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, ROOT)
dd = importlib.import_module("ngsf-hmm_amd.distributed")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
cap = 6 * 40
send = torch.zeros(cap, dtype=torch.float64)
recv = torch.full((world * cap,), float("nan"), dtype=torch.float64)
ex = dd.SiteExchange(send, recv, rank, world)
for k in (6, 6 * 7, cap):                 # one point, an odd number of points, the whole buffer
    send[:k] = 1000.0 * rank + torch.arange(k, dtype=torch.float64) + 0.5 * k
    ex(8 * k)
    want = torch.cat([1000.0 * r + torch.arange(k, dtype=torch.float64) + 0.5 * k for r in range(world)])
    assert torch.equal(recv[:world * k], want), (rank, k)
assert ex.calls == 3 and ex.bytes == 8 * (6 + 42 + cap)
for bad in (7, 8 * (cap + 6)):            # not whole doubles / more than the buffers hold
    try:
        ex(bad)
    except ValueError:
        pass
    else:
        raise SystemExit("an exchange that does not fit was accepted")
print("SITE_EXCHANGE_OK", rank, flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def test_site_exchange_two_ranks_gloo(tmp_path):
    """The all-gather a site-shard handle asks for, between two processes over gloo (CPU
    buffers): rank-major parts of every size, accounting, and sizes that do not fit refused."""
    world = 2
    script = tmp_path / "site_worker.py"
    script.write_text(f"ROOT = {ROOT!r}\n" + SITE_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    try:
        outs = [p.communicate(timeout=120)[0] for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0 and f"SITE_EXCHANGE_OK {r}" in outs[r], outs[r]


def test_site_ranges_ragged(pkg):
    dd = importlib.import_module("ngsf-hmm_amd.distributed")
    for S, V in ((1_000_000, 8), (1000, 3), (5000, 2), (33, 2), (100_003, 7)):
        rg = dd.site_ranges_ragged(S, V)
        assert rg[0][0] == 0 and rg[-1][1] == S and len(rg) == V
        assert all(a[1] == b[0] for a, b in zip(rg[:-1], rg[1:]))
        assert all(lo % 16 == 0 for lo, _ in rg)           # Viterbi back-pointer blocks stay whole
        sizes = [hi - lo for lo, hi in rg]
        assert min(sizes) > 0 and max(sizes) - min(sizes) <= 32
    with pytest.raises(ValueError):
        dd.site_ranges_ragged(3, 4)


def test_site_shard_algebra_against_the_oracle(pkg, orc_libm):
    """What the site-shard kernels compute, restated in numpy on a small case and held against
    the oracle's log-space recursions: a range of sites is the ordered product of its 2x2
    operators M_s = (c I + (1 - c) 1 q^T) diag(e_s); the row vector entering range r is
    q . prod_{r' < r} M_r', the column vector entering it from the right prod_{r' > r} M_r' . 1
    (k_fast_shard_edges); the log-likelihood is log(q . prod_r M_r . 1) (k_fast_shard_combine)
    and the posterior of a site the normalised product of the two vectors meeting there."""
    import orclib
    dd = importlib.import_module("ngsf-hmm_amd.distributed")
    I, S, V = 4, 330, 3
    d = pkg.simulate.simulate(I, S, seed=77, n_chrom=2, missing_rate=0.1, indF="r")
    gl = pkg.simulate.normalise_log_gl(d.gl)
    em = orclib.OracleEM(orc_libm, gl, d.pos_dist_mb)
    rng = np.random.default_rng(1)
    F, A = rng.uniform(0.05, 0.6, I), rng.uniform(0.05, 3.0, I)
    em.set_params(F, A, rng.uniform(0.05, 0.45, S))
    assert em.init_emission() == 0 and em.estep() == 0
    e = np.exp(em.e_prob)                                      # [I][S][2]
    ranges = dd.site_ranges_ragged(S, V)
    for i in range(I):
        q = np.array([1 - F[i], F[i]])

        def op(s):
            c = np.exp(-A[i] * d.pos_dist_mb[s])               # exp(-inf) = 0 at a chromosome start
            return (c * np.eye(2) + (1 - c) * np.outer(np.ones(2), q)) * e[i, s][None, :]
        M, scale = [], []
        for lo, hi in ranges:                                  # each range's operator, rescaled
            m, ex = np.eye(2), 0.0
            for s in range(lo, hi):
                m = m @ op(s)
                k = m.max()
                m, ex = m / k, ex + np.log(k)
            M.append(m)
            scale.append(ex)
        tot = np.eye(2)
        for m in M:
            tot = tot @ m
        lkl = np.log(q @ tot @ np.ones(2)) + sum(scale)
        np.testing.assert_allclose(lkl, em.ind_lkl[i], rtol=1e-12)
        for r, (lo, hi) in enumerate(ranges):
            u = q.copy()
            for m in M[:r]:
                u = u @ m
                u /= u.max()
            x = np.ones(2)
            for m in reversed(M[r + 1:]):
                x = m @ x
                x /= x.max()
            # forward vectors of the range from u, backward vectors from x
            fw = np.empty((hi - lo, 2))
            v = u
            for s in range(lo, hi):
                v = v @ op(s)
                v /= v.max()
                fw[s - lo] = v
            w = x
            for s in range(hi - 1, lo - 1, -1):
                p = fw[s - lo] * w
                p = p[1] / p.sum()
                p = 0.0 if p < 1e-5 else 1.0 if p > 1 - 1e-5 else p      # check_interv
                assert abs(p - em.marg[i, s]) < 1e-9, (i, s)
                w = op(s) @ w
                w /= w.max()


# ---- failure propagation between ranks: distributed.FailureBeacon ----
BEACON_WORKER = r'''
import importlib, os, sys, time
import torch
import torch.distributed as dist
from datetime import timedelta
sys.path.insert(0, ROOT)
pkg = importlib.import_module("ngsf-hmm_amd")
dd = importlib.import_module("ngsf-hmm_amd.distributed")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=300))
beacon = dd.make_beacon(rank, world, poll_s=0.1)
assert isinstance(beacon, dd.FailureBeacon)
part = torch.full((8,), float(rank), dtype=torch.float64)
whole = torch.empty(8 * world, dtype=torch.float64)
dist.all_gather_into_tensor(whole, part)        # one good exchange first
t0 = time.time()
try:
    with beacon.guard():
        if rank == FAILING:
            # a fatal of the reference on this rank's site range, between two collectives
            raise pkg.NgsFHMMError(-3, "invalid MAF!")
        dist.all_gather_into_tensor(whole, part)   # the failing rank never comes
        print("SURVIVOR_WAS_NOT_STOPPED", flush=True)
except pkg.NgsFHMMError as e:
    # (no sleep here: signal() itself lingers two polling intervals, so that the key outlives
    # a failing rank 0, in whose process the store lives -- bench.py exits the same way)
    print(f"RANK{rank}_RAISED {e}", flush=True)
    os._exit(4)
'''


@pytest.mark.parametrize("failing", [1, 0])
def test_a_failing_rank_ends_the_others_instead_of_hanging_them(tmp_path, failing):
    """One rank hits a fatal between two collectives (`invalid MAF!` on its own site range);
    the other sits in its next all-gather.  Without the beacon that wait ends with the process
    group's timeout (300 s here, 10 min under RCCL's watchdog, never under plain gloo); with it
    the waiting rank exits non-zero with the failing rank's message within seconds -- also
    when the failing rank is rank 0, which hosts the job's store: the key is read while the
    failing rank lingers, or, should the store die first, its loss is itself taken for a peer's
    failure."""
    import time
    world = 2
    script = tmp_path / "beacon_worker.py"
    script.write_text(f"ROOT = {ROOT!r}\nFAILING = {failing}\n" + BEACON_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29619 + failing), WORLD_SIZE=str(world))
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    try:
        outs = [p.communicate(timeout=120)[0] for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    other = 1 - failing
    assert time.time() - t0 < 60
    assert procs[failing].returncode == 4 and f"RANK{failing}_RAISED" in outs[failing], outs[failing]
    assert "invalid MAF!" in outs[failing]
    assert procs[other].returncode == 5, outs[other]
    assert "a peer failed -- rank" in outs[other], outs[other]
    assert (f"rank {failing}: NgsFHMMError" in outs[other] and "invalid MAF!" in outs[other]) or \
        "the job's store is lost" in outs[other], outs[other]
    assert "SURVIVOR_WAS_NOT_STOPPED" not in outs[other]


CLEAN_EXIT_WORKER = r'''
import importlib, os, sys, time
import torch
import torch.distributed as dist
from datetime import timedelta
sys.path.insert(0, ROOT)
dd = importlib.import_module("ngsf-hmm_amd.distributed")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=300))
beacon = dd.make_beacon(rank, world, poll_s=0.1)
part = torch.full((8,), float(rank), dtype=torch.float64)
whole = torch.empty(8 * world, dtype=torch.float64)
for _ in range(3):                               # the job's iterations, guarded as ShardedEM guards them
    with beacon.guard():
        dist.all_gather_into_tensor(whole, part)
if rank == 0:
    # rank 0 (the store's host) has nothing left to do and leaves, without ceremony
    print("RANK0_DONE", flush=True)
    os._exit(0)
# the other rank writes its shard's outputs for a while: the store is gone under its watcher
time.sleep(1.5)
print("RANK1_FINISHED_ITS_OUTPUTS", flush=True)
beacon.close()
os._exit(0)
'''


def test_a_rank_that_outlives_rank_zero_is_not_killed(tmp_path):
    """The c10d store lives in rank 0.  When rank 0 ends in good order a rank that is still writing
    its outputs finds the store gone: outside an EM iteration that is NOT a peer's failure (round-5
    advisory: the watcher used to end such a rank with exit code 5 and truncated outputs)."""
    world = 2
    script = tmp_path / "clean_exit_worker.py"
    script.write_text(f"ROOT = {ROOT!r}\n" + CLEAN_EXIT_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29641", WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    try:
        outs = [p.communicate(timeout=120)[0] for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert procs[0].returncode == 0 and "RANK0_DONE" in outs[0], outs[0]
    assert procs[1].returncode == 0 and "RANK1_FINISHED_ITS_OUTPUTS" in outs[1], outs[1]
    assert "a peer failed" not in outs[1], outs[1]


def test_bench_eight_rank_dry_launch():
    """`bench.py --gpus 8 --dry_launch`: the child launcher, the rendezvous of eight ranks, the
    known-answer preflight of the collectives and every rank's shard of the default workload in
    both layouts -- what an 8-GPU run does before its first GPU call, on the CPU (the build's
    one-GPU boxes admit at most six processes on a card -- four ranks and their launcher in practice
    -- so the eight-rank launch itself is rehearsed here).  The shards tile the 1000 x 1M job."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry_launch"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["dry_launch"] is True and line["n_gpus"] == 8 and line["preflight"]["world"] == 8
    ranks = sorted(line["ranks"], key=lambda x: x["rank"])
    assert [x["rank"] for x in ranks] == list(range(8))
    assert ranks[0]["sites"]["site_range"][0] == 0 and ranks[-1]["sites"]["site_range"][1] == 1_000_000
    assert all(a["sites"]["site_range"][1] == b["sites"]["site_range"][0] for a, b in zip(ranks, ranks[1:]))
    assert [x["individuals"]["ind_range"] for x in ranks] == [[125 * k, 125 * (k + 1)] for k in range(8)]
