"""Multi-GPU EM: individuals sharded over ranks, one process per GPU.

Everything per individual (E-step, L-BFGS-B M-step, Viterbi) needs no communication.
The allele-frequency step (est_maf, shared/gen_func.cpp:974-1009) needs, for a site,
the posteriors of EVERY individual, summed in individual order.  Each rank therefore
also owns a contiguous range of sites for that step and a static copy of all
individuals' genotype likelihoods for that range.  Per EM iteration:

  1. local E-step + indF/alpha M-step                        (no communication)
  2. all-to-all of the posteriors: rank r sends every other rank q the slice of its
     site-major posterior matrix that covers q's site range (contiguous, so no
     packing kernel is needed beyond one copy)                (RCCL all_to_all_single)
  3. est_maf on the own site range over all individuals, in global individual order
     (rank-major), i.e. exactly the single-GPU summation order
  4. all-gather of the per-range frequencies                  (RCCL all_gather)
  5. local emission refresh

This is the bit-faithful alternative to 101 all-reduce rounds (SURVEY.md section 8e,
design 2): two collectives per iteration, both using every xGMI link.

The compute backend is anything with the small interface of :class:`GpuBackend`
(tests/ uses a CPU stand-in with the gloo backend to exercise the exchange logic).
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np


def _staged(t):
    """gloo has no all-to-all / all-gather for device tensors: stage through the host.
    Only used when the process group is gloo (functional tests on one GPU); with the
    nccl (= RCCL) backend tensors stay on the device."""
    import torch.distributed as dist
    return dist.get_backend() == "gloo" and t.is_cuda


def all_to_all(recv, send):
    import torch.distributed as dist
    if _staged(send):
        r = recv.cpu()
        dist.all_to_all_single(r.view(-1), send.cpu().view(-1))
        recv.copy_(r)
    else:
        dist.all_to_all_single(recv.view(-1), send.view(-1))


def all_gather(out, inp):
    import torch.distributed as dist
    if _staged(inp):
        o = out.cpu()
        dist.all_gather_into_tensor(o, inp.cpu())
        out.copy_(o)
    else:
        dist.all_gather_into_tensor(out, inp)


def preflight(device=None):
    """Known-answer all-to-all and all-gather on float64 through the current process group,
    with the same calls the EM iteration makes (device tensors under nccl = RCCL; staged
    through the host under gloo).  Raises RuntimeError with the rank and the first wrong
    element: a collective that moves the wrong bytes must stop the run before any data is
    loaded, not bend the frequencies quietly."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    n = 257                                    # odd on purpose: not a multiple of anything
    # element j of the slice rank r sends to rank q
    send = torch.stack([1e6 * rank + 1e3 * q + torch.arange(n, dtype=torch.float64) / 1024.0
                        for q in range(world)]).to(dev)
    recv = torch.full_like(send, float("nan"))
    all_to_all(recv, send)
    part = (torch.arange(n, dtype=torch.float64) * (rank + 1) + 0.25 * rank).to(dev)
    whole = torch.full((world * n,), float("nan"), dtype=torch.float64, device=dev)
    all_gather(whole, part)
    if recv.is_cuda:
        torch.cuda.synchronize(recv.device)
    # (both collectives are issued before either is judged: every rank makes the same calls)
    want = torch.stack([1e6 * src + 1e3 * rank + torch.arange(n, dtype=torch.float64) / 1024.0
                        for src in range(world)])
    got = recv.cpu()
    if not torch.equal(got, want):
        bad = (got != want).nonzero()[0].tolist()
        raise RuntimeError(f"collective preflight: all_to_all_single delivered {got[tuple(bad)].item()!r} "
                           f"instead of {want[tuple(bad)].item()!r} at [source rank, element] {bad} "
                           f"on rank {rank} of {world} ({dist.get_backend()})")
    want = torch.cat([torch.arange(n, dtype=torch.float64) * (r + 1) + 0.25 * r for r in range(world)])
    got = whole.cpu()
    if not torch.equal(got, want):
        bad = int((got != want).nonzero()[0])
        raise RuntimeError(f"collective preflight: all_gather_into_tensor delivered {got[bad].item()!r} "
                           f"instead of {want[bad].item()!r} at element {bad} on rank {rank} of "
                           f"{world} ({dist.get_backend()})")
    return {"backend": dist.get_backend(), "world": world, "elements": n,
            "checked": ["all_to_all_single float64", "all_gather_into_tensor float64"]}


class FailureBeacon:
    """Failure propagation between the ranks of one job.

    The reference's fatal conditions (`invalid MAF!`, `invalid Lkl found!`, EM.cpp:400-410) are
    reachable from user data, and on several GPUs one of them can strike ONE rank -- its site
    range, its individuals -- between two collectives.  That rank leaves its iteration; the
    others would wait in their next all-gather / all-to-all for a partner that never comes (ten
    minutes under RCCL's watchdog, for ever under gloo).  The number of collectives left in an
    iteration is data dependent (the L-BFGS-B rounds), so a failing rank cannot keep the others
    company with poison values; instead the failure travels beside the data path:

    * a rank that fails writes ``<rank>: <message>`` under one key of the job's c10d store
      (``signal``) and raises its exception as usual;
    * every rank runs a daemon thread that polls that key (collectives release the GIL); when
      it appears the thread calls ``on_peer_failure(message)`` -- by default: the message on
      stderr and ``os._exit(5)``, a fresh exit (never an exec) that ends the waiting rank with a
      non-zero code and the failing rank's message within ``poll_s``.

    A store that no longer ANSWERS is a weaker sign than the key: the store lives in rank 0 (or
    the launcher), and a rank 0 that has finished in good order takes it along while another rank
    may still be writing its shard's outputs.  Its loss counts as a peer's failure only when it is
    seen on ``LOST_POLLS`` consecutive looks (a transient socket error is not one) AND this rank is
    inside a guarded block at each of them -- i.e. in an EM iteration, where a partner that is gone
    means a collective that never completes.  Outside (post-processing, shutdown) the watcher just
    stops.  Callers that go on working after their last iteration need nothing more; ``close()``
    before that work also stops the polling.

    Documented behaviour, not a recovery protocol: after a peer's fatal the job is over, as it
    is in the reference (`error()` + `exit(-1)`, gen_func.cpp:12-18)."""

    KEY = "nghmm_failed_rank"
    LOST_POLLS = 3

    def __init__(self, rank, poll_s=0.25, on_peer_failure=None, store=None):
        import threading
        import torch.distributed as dist
        self.rank = rank
        self.poll_s = poll_s
        self.on_peer_failure = on_peer_failure or self._exit
        base = store if store is not None else dist.distributed_c10d._get_default_store()
        self.store = dist.PrefixStore("nghmm_beacon", base)
        self._signalled = False
        self._busy = 0              # guarded blocks this rank is inside (its EM iterations)
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._watch, name="nghmm-failure-beacon", daemon=True)
        self._thread.start()

    def _exit(self, message):
        import os
        import sys
        sys.stderr.write(f"ngsf-hmm_amd: rank {self.rank} stops: a peer failed -- rank {message}\n")
        sys.stderr.flush()
        os._exit(5)

    def signal(self, message, linger=True):
        """This rank has failed: tell the others (idempotent).  linger: stay alive for two
        polling intervals afterwards -- the c10d store lives in rank 0's process, and a rank 0
        that wrote the key and exited at once would take the key with it before anybody read it."""
        if not self._signalled:
            self._signalled = True
            try:
                self.store.set(self.KEY, f"{self.rank}: {message}")
            except Exception:       # noqa: BLE001 - the store is gone: the job is ending anyway
                return
            if linger:
                import time
                time.sleep(2 * self.poll_s)

    def _watch(self):
        lost = 0
        while not self._stop.wait(self.poll_s):
            try:
                if not self.store.check([self.KEY]):
                    lost = 0
                    continue
                msg = self.store.get(self.KEY).decode(errors="replace")
            except Exception as e:  # noqa: BLE001
                # The store no longer answers.  After close() that is the job ending in good
                # order (process group destroyed); outside an iteration it is rank 0 having
                # finished before this rank -- nothing to act on.  INSIDE one, and repeatedly, the
                # process that hosts the store has died: under a launcher without an agent that
                # kills the survivors (mpirun, srun) they would sit in their next collective until
                # its timeout, so this too is a peer's failure.
                if self._signalled or self._stop.is_set():
                    return
                if self._busy <= 0:
                    return
                lost += 1
                if lost < self.LOST_POLLS:
                    continue
                self.on_peer_failure(f"0 (or the store's host): the job's store is lost "
                                     f"({type(e).__name__}: {e})")
                return
            if self._signalled or self._stop.is_set():
                return              # (our own failure: the main thread is raising it)
            self.on_peer_failure(msg)
            return

    def close(self):
        self._stop.set()

    def guard(self):
        """Context manager: an exception leaving the block is signalled to the peers first."""
        beacon = self

        class _G:
            def __enter__(self):
                beacon._busy += 1
                return beacon

            def __exit__(self, et, ev, tb):
                beacon._busy -= 1
                if ev is not None and not isinstance(ev, (KeyboardInterrupt, GeneratorExit)):
                    beacon.signal(f"{type(ev).__name__}: {ev}")
                return False
        return _G()


class _NoBeacon:
    def guard(self):
        import contextlib
        return contextlib.nullcontext()

    def signal(self, message):
        pass

    def close(self):
        pass


def make_beacon(rank, world, **kw):
    """A FailureBeacon when a multi-rank process group is up, else a no-op."""
    if world > 1:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return FailureBeacon(rank, **kw)
    return _NoBeacon()


def site_ranges(n_sites: int, world: int):
    """Contiguous, equal site ranges (the all-to-all uses equal splits)."""
    if n_sites % world != 0:
        raise ValueError(f"n_sites ({n_sites}) must be divisible by the number of ranks ({world})")
    step = n_sites // world
    return [(r * step, (r + 1) * step) for r in range(world)]


class GpuBackend:
    """The C-ABI handle plus torch tensors used as exchange buffers."""

    def __init__(self, pkg, n_ind, n_sites, device_index, mode):
        import torch
        self.torch = torch
        self.pkg = pkg
        self.hmm = pkg.NgsFHMM(n_ind, n_sites, device=device_index, mode=mode)
        self.device = torch.device("cuda", device_index)
        self.n_ind, self.n_sites = n_ind, n_sites

    def empty(self, *shape):
        return self.torch.empty(shape, device=self.device, dtype=self.torch.float64)

    def sync(self):
        """The library runs on its own HIP stream and the collectives on torch's: make the
        collective's result visible before the library reads it (and vice versa; library
        calls already synchronise their stream before returning)."""
        self.torch.cuda.synchronize(self.device)

    def load_device(self, gl, pos):
        self.hmm.load_device(gl.data_ptr(), pos.data_ptr())

    def load_chunks_device(self, pos, chunks, space=0, call_geno=False):
        """chunks: iterable of (site_begin, tensor [n][I][3]) on this device."""
        def feed():
            for s0, c in chunks:
                self.sync()                 # the library reads on its own stream
                yield s0, c.shape[0], c.data_ptr()
        # (`pos` too is a tensor torch may still be computing on ITS stream: the library copies it
        # on its own, non-blocking one as soon as the load begins -- before the first chunk's sync)
        self.sync()
        self.hmm.load_chunks_device(pos.data_ptr(), feed(), space=space, call_geno=call_geno)

    @property
    def packed(self):
        return bool(self.hmm.mode & self.pkg.GENO_PACKED)

    def geno_codes(self):
        """[S][I] uint8 genotype codes of a packed handle (device tensor)."""
        out = self.torch.empty((self.n_sites, self.n_ind), device=self.device,
                               dtype=self.torch.uint8)
        self.hmm._check(self.hmm.lib.nghmm_get_geno_codes_dev(self.hmm.handle, 0, self.n_sites,
                                                             C.c_void_p(out.data_ptr())))
        return out

    def load_geno_site_shard_device(self, codes):
        """codes: device tensor [S_own][I_tot] uint8."""
        self.hmm._check(self.hmm.lib.nghmm_load_geno_site_shard_dev(
            self.hmm.handle, C.c_void_p(codes.data_ptr())))

    def set_params(self, indF, alpha, freq):
        self.hmm.set_params(indF, alpha, freq)

    def init_emission(self):
        self.hmm.init_emission()

    def iter_em_local(self, freq_est, indF_fixed, alpha_fixed):
        st = self.hmm.iter_EM(freq_est, indF_fixed, alpha_fixed)
        return st, self.hmm.ind_lkl

    def estep(self):
        return self.hmm.estep()

    def mstep_indf(self, indF_fixed, alpha_fixed):
        return self.hmm.mstep_indf(indF_fixed, alpha_fixed)

    def estep_mstep(self, indF_fixed, alpha_fixed, after_estep):
        """E-step + indF/alpha M-step sharing their first pass over the emissions
        (nghmm_estep_mstep); after_estep() is called once the posteriors are final."""
        st = self.hmm.estep_mstep(indF_fixed, alpha_fixed, after_estep)
        return st, self.hmm.ind_lkl

    def pack_posteriors(self, lo, hi, out):
        self.hmm._check(self.hmm.lib.nghmm_pack_posteriors_dev(self.hmm.handle, lo, hi,
                                                               C.c_void_p(out.data_ptr())))

    def mstep_freq_sites(self, marg_blocks, freq_out):
        self.hmm._check(self.hmm.lib.nghmm_mstep_freq_sites_dev(
            self.hmm.handle, C.c_void_p(marg_blocks.data_ptr()), C.c_void_p(freq_out.data_ptr())))

    def set_freq(self, freq_all):
        self.hmm._check(self.hmm.lib.nghmm_set_freq_dev(self.hmm.handle,
                                                        C.c_void_p(freq_all.data_ptr())))

    def shard_config(self, n_ind_total, ind_begin, site_begin, n_sites_own):
        self.hmm._check(self.hmm.lib.nghmm_shard_config(self.hmm.handle, n_ind_total, ind_begin,
                                                        site_begin, n_sites_own))

    def load_site_shard_device(self, gl_shard):
        """gl_shard: device tensor [S_own][I_tot][3]."""
        self.hmm._check(self.hmm.lib.nghmm_load_gl_site_shard_dev(
            self.hmm.handle, C.c_void_p(gl_shard.data_ptr())))


class ShardedEM:
    """iter_EM across `world` ranks (world == 1: plain single-GPU path)."""

    def __init__(self, pkg, n_ind, n_sites, device_index=0, mode=None, rank=0, world=1,
                 backend=None, emulate_ranks=1):
        # emulate_ranks = V (world == 1 only): this process does the COMPUTE of rank 0 of a
        # V-rank run -- its n_ind individuals for all sites, and the frequency step on its
        # n_sites / V sites over V x n_ind individuals -- with the exchanges replaced by local
        # copies of the same size (the other ranks' posterior blocks and likelihood columns are
        # copies of its own; frequencies outside its site range keep their values).  For
        # predicting a rank's time on a one-GPU box; the results are not a cohort's.
        self.emulate = int(emulate_ranks) if world == 1 else 1
        if self.emulate > 1:
            world = self.emulate
        self.rank, self.world = rank, world
        self.n_ind, self.n_sites = n_ind, n_sites
        self.backend = backend or GpuBackend(pkg, n_ind, n_sites, device_index,
                                             pkg.MODE_FAST if mode is None else mode)
        self.hmm = getattr(self.backend, "hmm", None)
        self.ind_lkl = None
        # exchange accounting (milliseconds, summed over iterations; reset_timing() zeroes it):
        #   a2a_ms          duration of the posterior all-to-all (device events around it)
        #   a2a_exposed_ms  how long the host waited for it after the objective rounds were done
        #   allgather_ms    the frequency all-gather (host clock, synchronous)
        #   freq_step_ms    est_maf on the own site range (host clock)
        self.timing = dict(a2a_ms=0.0, a2a_exposed_ms=0.0, allgather_ms=0.0, freq_step_ms=0.0,
                           iterations=0)
        self._ev = None
        self.beacon = make_beacon(rank, world if self.emulate == 1 else 1)
        if world > 1:
            self.ranges = site_ranges(n_sites, world)
            lo, hi = self.ranges[rank]
            self.S_own = hi - lo
            self.backend.shard_config(n_ind * world, rank * n_ind, lo, self.S_own)
            self._send = self.backend.empty(world, self.S_own, n_ind)
            self._recv = self.backend.empty(world, self.S_own, n_ind)
            self._freq_own = self.backend.empty(self.S_own)
            self._freq_all = self.backend.empty(n_sites)

    # -- data ---------------------------------------------------------------
    def load_device(self, gl, pos):
        """gl: tensor [S][I_local][3]; pos: tensor [S] (on the backend's device)."""
        self.backend.load_device(gl, pos)
        if self.emulate > 1:
            lo, hi = self.ranges[self.rank]
            shard = gl[lo:hi].repeat(1, self.world, 1).contiguous()   # [S_own][V I][3]
            self._sync()
            self.backend.load_site_shard_device(shard)
            del shard
        elif self.world > 1:
            self._exchange_site_shard(gl)

    def load_chunks_device(self, pos, chunks, space=0, call_geno=False):
        """Chunked loading from device tensors (site_begin, [n][I_local][3]); with several
        ranks the handle must be packed: the site shards are then built from the genotype
        codes (a dense handle's shard needs the whole matrix: use load_device)."""
        self.backend.load_chunks_device(pos, chunks, space=space, call_geno=call_geno)
        if self.emulate > 1:
            raise ValueError("emulate_ranks works on likelihood data loaded with load_device")
        if self.world > 1:
            if not getattr(self.backend, "packed", False):
                raise ValueError("chunked loading on several ranks needs a packed handle")
            self._exchange_code_shard()

    def _exchange_code_shard(self):
        """One-off, packed handles: every rank's [S][I_loc] code bytes -> [S_own][I_tot]."""
        world, I = self.world, self.n_ind
        codes = self.backend.geno_codes()                          # [S][I] uint8
        send = codes.reshape(world, self.S_own, I).contiguous()
        recv = send.new_empty((world, self.S_own, I))
        self._sync()
        all_to_all(recv, send)
        shard = recv.permute(1, 0, 2).contiguous().view(self.S_own, world * I)
        self._sync()
        self.backend.load_geno_site_shard_device(shard)
        del codes, send, recv, shard

    def _exchange_site_shard(self, gl):
        """One-off: build [S_own][I_tot][3] from every rank's [S][I_loc][3]."""
        import torch.distributed as dist
        world, I = self.world, self.n_ind
        send = gl.reshape(world, self.S_own, I, 3).contiguous()   # slices by destination
        recv = self.backend.empty(world, self.S_own, I, 3)
        all_to_all(recv, send)
        shard = recv.permute(1, 0, 2, 3).contiguous().view(self.S_own, world * I, 3)
        self._sync()
        self.backend.load_site_shard_device(shard)
        del send, recv, shard

    def set_params(self, indF, alpha, freq):
        self.backend.set_params(indF, alpha, freq)

    def init_emission(self):
        self.backend.init_emission()

    # -- one EM iteration ------------------------------------------------------
    def iter_EM(self, freq_est=1, indF_fixed=False, alpha_fixed=False):
        """One EM iteration of the cohort.  A failure of this rank (the reference's fatals are
        reachable from user data) is signalled to the others before it is raised
        (:class:`FailureBeacon`): they end with its message instead of waiting in a collective."""
        with self.beacon.guard():
            return self._iter_EM(freq_est, indF_fixed, alpha_fixed)

    def _iter_EM(self, freq_est, indF_fixed, alpha_fixed):
        if self.emulate > 1:
            return self._iter_em_emulated(freq_est, indF_fixed, alpha_fixed)
        if self.world == 1:
            st, self.ind_lkl = self.backend.iter_em_local(freq_est, indF_fixed, alpha_fixed)
            return st
        # the posteriors are final after the E-step and the indF/alpha M-step neither reads
        # nor writes them: start their all-to-all then and let it run under the M-step
        box = {}

        def after_estep():
            if freq_est:
                box["work"] = self.start_posterior_exchange()

        fused = getattr(self.backend, "estep_mstep", None)
        if fused is not None:
            st, self.ind_lkl = fused(indF_fixed, alpha_fixed, after_estep)
        else:
            self.ind_lkl = self.backend.estep()
            after_estep()
            st = self.backend.mstep_indf(indF_fixed, alpha_fixed)
        if freq_est:
            self.finish_exchange_and_update_freq(box.get("work"))
        return st

    def reset_timing(self):
        for k in self.timing:
            self.timing[k] = 0 if k == "iterations" else 0.0

    def collective_bytes_per_iter(self):
        """Bytes that leave this rank's GPU per EM iteration: the posterior slices of the other
        ranks' site ranges, and its own frequencies to everybody (ring all-gather: each rank
        forwards world - 1 pieces)."""
        if self.world == 1:
            return {"all_to_all_out": 0, "all_gather_out": 0}
        w = self.world
        return {"all_to_all_out": 8 * self.S_own * self.n_ind * (w - 1),
                "all_gather_out": 8 * self.S_own * (w - 1)}

    def _events(self):
        torch = getattr(self.backend, "torch", None)
        if torch is None or not self._send.is_cuda:
            return None
        if self._ev is None:
            self._ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        return self._ev

    def _iter_em_emulated(self, freq_est, indF_fixed, alpha_fixed):
        """Rank 0's kernels of a `world`-rank iteration; every exchange a local copy."""
        def after_estep():
            if freq_est:
                self.backend.pack_posteriors(0, self.n_sites, self._send)
                self._sync()
                for q in range(self.world):        # "receive" a block from every rank
                    self._recv[q].copy_(self._send[self.rank])
        st, self.ind_lkl = self.backend.estep_mstep(indF_fixed, alpha_fixed, after_estep)
        if freq_est:
            self._sync()
            self.backend.mstep_freq_sites(self._recv, self._freq_own)
            lo, hi = self.ranges[self.rank]
            if not getattr(self, "_freq_all_filled", False):   # the other ranges keep these values
                self._freq_all.copy_(self.backend.torch.from_numpy(self.hmm.freq).to(self._freq_all.device))
                self._freq_all_filled = True
            self._freq_all[lo:hi] = self._freq_own
            self._sync()
            self.backend.set_freq(self._freq_all)
        return st

    def start_posterior_exchange(self):
        import torch.distributed as dist
        # equal contiguous site ranges: the send buffer [rank][S_own][I] is the whole
        # site-major posterior matrix
        self.backend.pack_posteriors(0, self.n_sites, self._send)
        self._sync()
        ev = self._events()
        if ev:
            ev[0].record()
        if _staged(self._send):
            # gloo through the host: synchronous, nothing of it is hidden
            import time
            t0 = time.perf_counter()
            all_to_all(self._recv, self._send)
            self.timing["a2a_exposed_ms"] += (time.perf_counter() - t0) * 1e3
            if ev:
                ev[1].record()
            return None
        return dist.all_to_all_single(self._recv.view(-1), self._send.view(-1), async_op=True)

    def finish_exchange_and_update_freq(self, work):
        import time
        t0 = time.perf_counter()
        ev = self._events()
        if work is not None:
            work.wait()            # torch's current stream now waits for the collective
            if ev:
                ev[1].record()
        self._sync()
        t1 = time.perf_counter()
        # _recv is [source rank][S_own][I_loc]: the rank-blocked layout est_maf reads
        self.backend.mstep_freq_sites(self._recv, self._freq_own)
        t2 = time.perf_counter()
        all_gather(self._freq_all, self._freq_own)
        self._sync()
        t3 = time.perf_counter()
        self.backend.set_freq(self._freq_all)
        tm = self.timing
        if work is not None:
            tm["a2a_exposed_ms"] += (t1 - t0) * 1e3
        tm["freq_step_ms"] += (t2 - t1) * 1e3
        tm["allgather_ms"] += (t3 - t2) * 1e3
        if ev:
            tm["a2a_ms"] += ev[0].elapsed_time(ev[1])
        tm["iterations"] += 1

    def _sync(self):
        if hasattr(self.backend, "sync"):
            self.backend.sync()

    def close(self):
        self.beacon.close()
        if self.hmm is not None:
            self.hmm.close()
        elif hasattr(self.backend, "close"):
            self.backend.close()


# ---------------------------------------------------------------------------------------------
# Fast mode: shard the SITES (include/nghmm.h, "shard the SITES instead")
# ---------------------------------------------------------------------------------------------
# A run of sites is a product of 2x2 operators, so in fast mode the site axis can be cut between
# GPUs the way it is cut between lane-chunks inside one: every rank holds ALL individuals for a
# contiguous range of sites, and the ranges owe each other six doubles per individual and E-step
# and per objective point and round -- a few hundred KB through one small all-gather, where the
# individual shards above move every posterior (8 GB per iteration at 1000 x 1M, and between
# two GPUs over ONE xGMI link).  The frequency step has every individual of its sites at hand
# and exchanges nothing; every rank runs the same L-BFGS-B steps for all individuals from the
# same gathered bits.


def site_ranges_ragged(n_sites: int, world: int):
    """Contiguous site ranges of near-equal length (multiples of 16 sites except the last, so
    that the blocked Viterbi back-pointers of neighbouring ranges do not share a block)."""
    if world < 1 or n_sites < world:
        raise ValueError(f"{n_sites} sites over {world} ranks")
    cuts = [0]
    for r in range(1, world):
        c = (n_sites * r // world) // 16 * 16
        cuts.append(max(c, cuts[-1] + 1))
    cuts.append(n_sites)
    if any(b <= a for a, b in zip(cuts[:-1], cuts[1:])):
        raise ValueError(f"{n_sites} sites are too few for {world} ranks")
    return list(zip(cuts[:-1], cuts[1:]))


class SiteExchange:
    """The all-gather a site-shard handle asks for (nghmm_allgather_fn): the first n bytes of
    `send` of every rank into `recv` = [rank][n], ordered on the handle's HIP stream.

    * process group nccl (= RCCL), device buffers: `all_gather_into_tensor` issued with the
      handle's stream as torch's current stream -- the collective waits for what the library
      enqueued before, the library's next kernels wait for the collective; the host waits for
      nothing;
    * process group gloo: through the host, synchronous (CPU buffers directly);
    * `emulate` (one rank's compute of a V-rank run on a one-GPU box): V copies of the own
      part on the handle's stream; with `through_group` the part first goes through
      `all_gather_into_tensor` of the current (one-rank) process group, issued exactly as
      above -- the nccl code path and its per-call cost on a box with one GPU."""

    def __init__(self, send, recv, rank, world, stream_ptr=None, emulate=False, through_group=False):
        self.send, self.recv = send, recv
        self.rank, self.world = rank, world
        self.emulate = emulate
        self.through_group = through_group
        self._tmp = send.new_empty(send.numel()) if through_group else None
        self.calls = 0
        self.bytes = 0
        self.host_ms = 0.0
        # every call's duration, for the line a multi-GPU run prints about itself: under nccl a
        # pair of events on the handle's stream around the collective (it starts when this
        # rank's part is ready and ends when the slowest rank's has arrived: duration = wire time
        # + waiting for the others), otherwise the host's clock; `take_log()` reads and clears
        self._log = []
        # (the pair of events is two more packets on the stream per call: a caller that times a loop
        # clears `timed` and sets it for the iterations it wants to see)
        self.timed = True
        self._stream = None
        if send.is_cuda and stream_ptr is not None:
            import torch
            self._stream = torch.cuda.ExternalStream(stream_ptr, device=send.device)

    def __call__(self, n_bytes):
        import time
        import torch
        t0 = time.perf_counter()
        if n_bytes % 8 or n_bytes * self.world > self.recv.numel() * 8:
            raise ValueError(f"site-shard exchange of {n_bytes} bytes does not fit the buffers")
        k = n_bytes // 8
        part, whole = self.send[:k], self.recv[:k * self.world]
        if self.emulate:
            with torch.cuda.stream(self._stream):
                if self.through_group:
                    import torch.distributed as dist
                    dist.all_gather_into_tensor(self._tmp[:k], part)
                    part = self._tmp[:k]
                whole.view(self.world, k).copy_(part.expand(self.world, k))
        else:
            import torch.distributed as dist
            if dist.get_backend() == "gloo" and part.is_cuda:
                torch.cuda.synchronize(part.device)      # the library's stream has written `send`
                w = torch.empty(k * self.world, dtype=part.dtype)
                dist.all_gather_into_tensor(w, part.cpu())
                whole.copy_(w)
                torch.cuda.synchronize(part.device)
            elif part.is_cuda:
                with torch.cuda.stream(self._stream):
                    if self.timed:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        dist.all_gather_into_tensor(whole, part)
                        e1.record()
                        if len(self._log) < 65536:
                            self._log.append((e0, e1))
                    else:
                        dist.all_gather_into_tensor(whole, part)
                self.calls += 1
                self.bytes += n_bytes
                self.host_ms += (time.perf_counter() - t0) * 1e3
                return
            else:
                dist.all_gather_into_tensor(whole, part)
        self.calls += 1
        self.bytes += n_bytes
        dt = (time.perf_counter() - t0) * 1e3
        self.host_ms += dt
        if len(self._log) < 65536:
            self._log.append(dt)

    def take_log(self):
        """Milliseconds of every call since the last take_log() (call after the stream has
        drained: event pairs are read here)."""
        out = []
        for x in self._log:
            out.append(x if isinstance(x, float) else float(x[0].elapsed_time(x[1])))
        self._log = []
        return out


class SiteShardedEM:
    """iter_EM over `world` ranks, rank r holding all `n_ind` individuals for the site range
    site_ranges_ragged(n_sites, world)[r]; NGHMM_MODE_FAST (dense or packed).  world == 1 with
    emulate_ranks = V: rank 0's compute of a V-rank run, the exchanges V local copies."""

    def __init__(self, pkg, n_ind, n_sites, device_index=0, mode=None, rank=0, world=1,
                 emulate_ranks=1, emulate_through_group=False):
        import torch
        self.emulate = int(emulate_ranks) if world == 1 else 1
        if self.emulate > 1:
            world = self.emulate
        self.pkg = pkg
        self.rank, self.world = rank, world
        self.n_ind, self.n_sites = n_ind, n_sites
        self.ranges = site_ranges_ragged(n_sites, world)
        self.site_lo, self.site_hi = self.ranges[rank]
        self.S_own = self.site_hi - self.site_lo
        mode = pkg.MODE_FAST if mode is None else mode
        if (mode & 3) != pkg.MODE_FAST:
            raise ValueError("site shards are a fast-mode layout (exact mode: ShardedEM)")
        self.hmm = pkg.NgsFHMM(n_ind, self.S_own, device=device_index, mode=mode)
        self.device = torch.device("cuda", device_index)
        self.ind_lkl = None
        self.exchange = None
        self.timing = dict(iterations=0)
        self.beacon = make_beacon(rank, world if self.emulate == 1 else 1)
        if world > 1:
            nbytes = self.hmm.site_shard_bytes()
            self._send = torch.zeros(nbytes // 8, dtype=torch.float64, device=self.device)
            self._recv = torch.zeros(world * (nbytes // 8), dtype=torch.float64, device=self.device)
            torch.cuda.synchronize(self.device)
            self.exchange = SiteExchange(self._send, self._recv, rank, world,
                                         stream_ptr=self.hmm.lib.nghmm_stream(self.hmm.handle),
                                         emulate=self.emulate > 1,
                                         through_group=self.emulate > 1 and emulate_through_group)
            self.hmm.site_shard_setup(rank, world, self._send.data_ptr(), self._recv.data_ptr(),
                                      nbytes, self.exchange)

    # -- data: the rank's own site range, all individuals ------------------------------------
    def load_device(self, gl, pos):
        """gl: device tensor [S_own][n_ind][3]; pos: [S_own] distances in Mb, the first one the
        true distance to the range before (+inf only at a chromosome start)."""
        import torch
        torch.cuda.synchronize(self.device)
        self.hmm.load_device(gl.data_ptr(), pos.data_ptr())

    def load_chunks_device(self, pos, chunks, space=0, call_geno=False):
        def feed():
            import torch
            for lo, c in chunks:          # one chunk alive at a time
                torch.cuda.synchronize(self.device)   # the library reads on its own stream
                yield lo, c.shape[0], c.data_ptr()
        import torch
        torch.cuda.synchronize(self.device)   # `pos`: copied as the load begins, before the first chunk
        self.hmm.load_chunks_device(pos.data_ptr(), feed(), space=space, call_geno=call_geno)

    def set_params(self, indF, alpha, freq):
        self.hmm.set_params(indF, alpha, freq)

    def init_emission(self):
        self.hmm.init_emission()

    def iter_EM(self, freq_est=1, indF_fixed=False, alpha_fixed=False):
        """One EM iteration of the chain.  The intended --freq_est 2 walks the sites in order on
        ONE handle and is refused here (as nghmm_chain_iter_em refuses it).  A failure of this
        rank -- a fatal of the reference on its site range, an exception in the exchange -- is
        signalled to the other ranks before it is raised (:class:`FailureBeacon`)."""
        if self.world > 1 and (int(freq_est) & (self.pkg.LD_INTENDED | self.pkg.EPROB_LD)):
            raise self.pkg.NgsFHMMError(-10, "the intended --freq_est 2 walks the sites in order on "
                                             "ONE handle: not available for a chain of site shards")
        with self.beacon.guard():
            st = self.hmm.iter_EM(freq_est, indF_fixed, alpha_fixed)
        self.ind_lkl = self.hmm.ind_lkl          # the chain's: the same on every rank
        self.timing["iterations"] += 1
        return st

    def reset_timing(self):
        self.timing["iterations"] = 0
        if self.exchange:
            self.exchange.calls = self.exchange.bytes = 0
            self.exchange.host_ms = 0.0

    def collective_bytes_per_iter(self):
        """Bytes that leave this rank's GPU per EM iteration (measured: the own part of every
        all-gather, to world - 1 ranks)."""
        it = max(self.timing["iterations"], 1)
        if not self.exchange:
            return {"all_to_all_out": 0, "all_gather_out": 0, "all_gathers": 0}
        return {"all_to_all_out": 0,
                "all_gather_out": self.exchange.bytes * (self.world - 1) / it,
                "all_gathers": self.exchange.calls / it}

    # -- results ------------------------------------------------------------------------------
    def _bcast(self, a, src):
        """numpy array of rank `src` to every rank."""
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(np.ascontiguousarray(a))
        if dist.get_backend() == "nccl":
            t = t.to(self.device)
        dist.broadcast(t, src=src)
        return t.cpu().numpy()

    def viterbi(self):
        """Most probable path of the own site range, [n_ind][S_own]: the forward halves in rank
        order, the backward halves in reverse order, the boundary scores / states by broadcast
        (include/nghmm.h, nghmm_viterbi_shard_forward / _back)."""
        if self.emulate > 1:
            raise ValueError("emulate_ranks has no chain to decode")
        if self.world == 1:
            return self.hmm.viterbi()
        with self.beacon.guard():
            return self._viterbi_chain()

    def _viterbi_chain(self):
        scores = None
        for r in range(self.world):
            out = self.hmm.viterbi_shard_forward(scores) if r == self.rank else \
                np.empty((self.n_ind, 2))
            got = self._bcast(out, r)
            if r + 1 == self.rank:
                scores = got
        state, path = None, None
        for r in reversed(range(self.world)):
            if r == self.rank:
                before, path = self.hmm.viterbi_shard_back(state)
            else:
                before = np.empty(self.n_ind, dtype=np.uint8)
            got = self._bcast(before, r)
            if r - 1 == self.rank:
                state = got
        return path

    def gather_freq(self):
        """freq of all sites on every rank (the .indF file's frequency block)."""
        import torch
        import torch.distributed as dist
        mine = np.ascontiguousarray(self.hmm.freq)
        if self.world == 1 or self.emulate > 1:
            return mine
        parts = []
        for r, (lo, hi) in enumerate(self.ranges):
            parts.append(self._bcast(mine if r == self.rank else np.empty(hi - lo), r))
        return np.concatenate(parts)

    def close(self):
        self.beacon.close()
        if self.hmm is not None:
            self.hmm.close()
            self.hmm = None
