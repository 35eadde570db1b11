"""Synthetic ngsF-HMM inputs.

Restates the data model of the reference's simulator, scripts/ngsF-HMMsim.R
(R is not installed here, and its RNG stream cannot be reproduced anyway):

* site distances ``d_s = max(1, int(N(1e5, (1e5/3)^2)))`` bp, cumulative positions
  on one chromosome (ngsF-HMMsim.R:192-196); several chromosomes restart the
  coordinate, which the reader turns into an infinite distance
  (shared/read_data.cpp:203-210);
* per-individual IBD path: first state ~ Bernoulli(F), then a Markov chain with
  ``P(0->1) = (1-e^{-a d})F``, ``P(1->0) = (1-e^{-a d})(1-F)``, d in Mb
  (ngsF-HMMsim.R:22-47);
* two haplotypes ~ Bernoulli(freq) per site, the first copied from the second
  where IBD, genotype = sum (ngsF-HMMsim.R:238-247);
* reads: depth ~ Poisson(depth), minor-allele reads ~ Binomial(depth,
  {e, 0.5, 1-e}[g]); GL = the three binomial likelihoods normalised to sum 1,
  natural log, rounded to 10 decimals (ngsF-HMMsim.R:48-67).

The generator is numpy's PCG64 with a fixed seed, so fixtures are reproducible.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np


@dataclass
class SimData:
    n_ind: int
    n_sites: int
    gl: np.ndarray          # [S][I][3] float64 natural-log GL (normalised), file order
    geno: np.ndarray        # [S][I] int8 true genotypes
    path: np.ndarray        # [I][S] uint8 true IBD states
    chrom: np.ndarray       # [S] int32 chromosome index
    pos: np.ndarray         # [S] int64 position in bp
    pos_dist_mb: np.ndarray  # [S] float64 distance to previous site in Mb; inf at chr starts
    freq: np.ndarray        # [S] true allele frequencies
    indF: np.ndarray        # [I]
    alpha: np.ndarray       # [I]


def reader_distances(chrom: np.ndarray, pos: np.ndarray) -> np.ndarray:
    """Distances as shared/read_data.cpp:165-218 + ngsF-HMM.cpp:75-86 derive them.

    The first site's distance is its absolute position (prev_pos starts at 0); a
    chromosome change gives +inf; everything is divided by 1e6 (Mb).
    """
    S = len(pos)
    d = np.empty(S, dtype=np.float64)
    prev_pos = 0
    prev_chr = chrom[0] if S else 0
    for s in range(S):
        if chrom[s] == prev_chr:
            d[s] = float(pos[s]) - float(prev_pos)
        else:
            d[s] = math.inf
            prev_chr = chrom[s]
        prev_pos = int(pos[s])
    return d / 1e6


def simulate(n_ind: int, n_sites: int, *, freq=0.2, indF=0.5, alpha=0.01, depth=2.0,
             error=0.01, seed=12345, n_chrom: int = 1, missing_rate: float = 0.0) -> SimData:
    """Generate one data set.  ``freq``/``indF``/``alpha`` may be a float or ``"r"``
    (uniform(0,1), ngsF-HMMsim.R:108-148)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    I, S = n_ind, n_sites

    F = rng.random(I) if indF == "r" else np.full(I, float(indF))
    A = rng.random(I) if alpha == "r" else np.full(I, float(alpha))
    fr = rng.random(S) if freq == "r" else np.full(S, float(freq))

    # positions
    gaps = rng.normal(1e5, 1e5 / 3.0, size=S).astype(np.int64)
    gaps[gaps < 1] = 1
    per_chr = -(-S // n_chrom)
    chrom = (np.arange(S) // per_chr).astype(np.int32)
    pos = np.empty(S, dtype=np.int64)
    for c in range(n_chrom):
        sel = chrom == c
        pos[sel] = np.cumsum(gaps[sel])
    pos_dist_mb = reader_distances(chrom, pos)
    # the simulator itself uses the raw gaps (in Mb) for the path; chr starts restart the chain
    gap_mb = gaps.astype(np.float64) / 1e6
    gap_mb[np.isinf(pos_dist_mb)] = np.inf

    # IBD paths.  P(0->1) = (1-X)F and P(1->0) = (1-X)(1-F) is the chain "with
    # probability 1-X redraw the state from Bernoulli(F), else keep it", which
    # vectorises: the state at s is the draw made at the last redraw site <= s.
    X = np.exp(-A[:, None] * gap_mb[None, :])          # [I][S]; 0 at chromosome starts
    redraw = rng.random((I, S)) >= X
    redraw[:, 0] = True
    draws = (rng.random((I, S)) < F[:, None]).astype(np.uint8)
    last = np.where(redraw, np.arange(S)[None, :], 0)
    np.maximum.accumulate(last, axis=1, out=last)
    path = np.take_along_axis(draws, last, axis=1)
    del X, redraw, draws, last

    # genotypes
    hap1 = (rng.random((I, S)) < fr[None, :]).astype(np.int8)
    hap2 = (rng.random((I, S)) < fr[None, :]).astype(np.int8)
    hap1[path == 1] = hap2[path == 1]
    geno_is = hap1 + hap2  # [I][S]

    # genotype likelihoods
    dep = rng.poisson(depth, size=(I, S))
    if missing_rate > 0:
        dep[rng.random((I, S)) < missing_rate] = 0
    p_read = np.array([error, 0.5, 1.0 - error])
    nA = rng.binomial(dep, p_read[geno_is])
    # log binomial likelihoods up to the common binomial coefficient, then normalise
    with np.errstate(divide="ignore"):
        lp = np.log(p_read)
        lq = np.log1p(-p_read)
    ll = nA[..., None] * lp[None, None, :] + (dep - nA)[..., None] * lq[None, None, :]
    m = ll.max(axis=2, keepdims=True)
    lse = m + np.log(np.exp(ll - m).sum(axis=2, keepdims=True))
    gl_is = np.round(ll - lse, 10)  # [I][S][3]

    gl = np.ascontiguousarray(np.transpose(gl_is, (1, 0, 2)))  # [S][I][3]
    geno = np.ascontiguousarray(geno_is.T).astype(np.int8)
    return SimData(I, S, gl, geno, path, chrom, pos, pos_dist_mb, fr, F, A)


def normalise_log_gl(gl: np.ndarray) -> np.ndarray:
    """post_prob(gl, gl, NULL) applied once (shared/gen_func.cpp:920-932), vectorised:
    subtract the log-sum-exp of the three values.  Host-side convenience for tests
    and the bench; bit-level agreement with the reference's double application
    (read_data.cpp:40 and ngsF-HMM.cpp:116) is not claimed here."""
    m = gl.max(axis=-1, keepdims=True)
    lse = m + np.log(np.exp(gl - m).sum(axis=-1, keepdims=True))
    return gl - lse


def called_genotype_gl(geno: np.ndarray) -> np.ndarray:
    """Log GLs the reference builds from called genotypes {-1,0,1,2}
    (shared/read_data.cpp:88-98): one-hot with -1e15 elsewhere, -1 = uniform."""
    S, I = geno.shape
    out = np.full((S, I, 3), -1e15, dtype=np.float64)
    for g in range(3):
        out[..., g][geno == g] = 0.0
    out[geno < 0] = math.log(1.0 / 3.0)
    return normalise_log_gl(out)


def simulate_torch_chunks(n_ind: int, n_sites: int, device, *, freq=0.2, indF=0.5, alpha=0.01,
                          depth=2.0, error=0.01, seed=12345, chunk_sites: int = 20000,
                          pos_seed=None, n_chrom: int = 1):
    """The same data model generated directly on a GPU with torch, a chunk of sites at a time.

    Returns (pos_dist_mb [S] device tensor, generator of (site_begin, gl_chunk [n][I][3]))
    with gl_chunk float64 normalised natural-log GL.  n_chrom > 1: equal runs of sites whose
    first distance is +inf, as the reader derives it at a chromosome change
    (shared/read_data.cpp:203-210).  Not bit-compatible with :func:`simulate` (different
    RNG); same distributions.
    """
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    I, S = n_ind, n_sites
    f64 = torch.float64
    # pos_seed: the site positions (and site frequencies) from a generator of their own, so
    # that the ranks of a multi-GPU run, which simulate different individuals (seed), share
    # one set of sites
    gs = g
    if pos_seed is not None:
        gs = torch.Generator(device=device)
        gs.manual_seed(int(pos_seed))
    gaps = torch.normal(1e5, 1e5 / 3.0, (S,), generator=gs, device=device, dtype=f64).to(torch.int64)
    gaps.clamp_(min=1)
    pos_dist_mb = gaps.to(f64) / 1e6     # cumulative positions: d_0 = pos_0 - 0
    if n_chrom > 1:
        per_chr = -(-S // n_chrom)
        pos_dist_mb[per_chr::per_chr] = float("inf")
    p_read = torch.tensor([error, 0.5, 1.0 - error], device=device, dtype=f64)
    lp, lq = torch.log(p_read), torch.log1p(-p_read)
    state0 = (torch.rand((I,), generator=g, device=device) < indF).to(torch.int64)
    # freq = "r": a uniform frequency per site (ngsF-HMMsim.R:127-133), else one value
    site_freq = (torch.rand((S,), generator=gs, device=device, dtype=f64) if isinstance(freq, str)
                 else torch.full((S,), float(freq), device=device, dtype=f64))

    def chunks():
        state = state0
        first = True
        for s0 in range(0, S, chunk_sites):
            s1 = min(S, s0 + chunk_sites)
            n = s1 - s0
            fr = site_freq[None, s0:s1]
            X = torch.exp(-alpha * pos_dist_mb[s0:s1])                       # [n]; 0 at chr starts
            redraw = torch.rand((I, n), generator=g, device=device, dtype=f64) >= X[None, :]
            if first:
                redraw[:, 0] = True
                first = False
            draws = (torch.rand((I, n), generator=g, device=device) < indF).to(torch.int64)
            idx = torch.where(redraw, torch.arange(n, device=device)[None, :], -1)
            idx = torch.cummax(idx, dim=1).values
            path = torch.where(idx >= 0, torch.gather(draws, 1, idx.clamp(min=0)), state[:, None])
            state = path[:, -1].clone()
            h1 = (torch.rand((I, n), generator=g, device=device) < fr).to(torch.int64)
            h2 = (torch.rand((I, n), generator=g, device=device) < fr).to(torch.int64)
            h1 = torch.where(path == 1, h2, h1)
            geno = h1 + h2                                                    # [I][n]
            dep = torch.poisson(torch.full((I, n), float(depth), device=device, dtype=f64),
                                generator=g)
            nA = torch.binomial(dep, p_read[geno], generator=g)
            ll = nA[..., None] * lp + (dep - nA)[..., None] * lq             # [I][n][3]
            ll = ll - torch.logsumexp(ll, dim=2, keepdim=True)
            ll = torch.round(ll, decimals=10)
            ll = ll - torch.logsumexp(ll, dim=2, keepdim=True)               # the reader's post_prob
            out = ll.permute(1, 0, 2).contiguous()
            del X, redraw, draws, idx, path, h1, h2, geno, dep, nA, ll
            yield s0, out

    return pos_dist_mb, chunks()


def simulate_torch(n_ind: int, n_sites: int, device, *, freq=0.2, indF=0.5, alpha=0.01,
                   depth=2.0, error=0.01, seed=12345, chunk_sites: int = 20000, pos_seed=None,
                   n_chrom: int = 1):
    """:func:`simulate_torch_chunks` collected into one tensor, for benchmark-sized inputs
    (1000 x 1M = 24 GB of doubles never touches the host).  Returns (gl [S][I][3],
    pos_dist_mb [S]) as device tensors."""
    import torch

    pos_dist_mb, chunks = simulate_torch_chunks(n_ind, n_sites, device, freq=freq, indF=indF,
                                                alpha=alpha, depth=depth, error=error, seed=seed,
                                                chunk_sites=chunk_sites, pos_seed=pos_seed,
                                                n_chrom=n_chrom)
    gl = torch.empty((n_sites, n_ind, 3), device=device, dtype=torch.float64)
    for s0, c in chunks:
        gl[s0:s0 + c.shape[0]] = c
        del c
    return gl, pos_dist_mb


# ---------------------------------------------------------------------------------------------
# Index-addressed generator: every random number is a hash of (seed, stream, individual, site)
# ---------------------------------------------------------------------------------------------
# A multi-GPU job must process the data set the one-GPU job processes (results do not depend on
# the number of workers: EM.cpp:151-161,198-201), whatever the sharding.  The generators above
# draw from a sequential stream, so what a rank gets depends on the shapes it asks for; here
# element (i, s) of every random field is a counter-based hash of its GLOBAL indices, every
# operation after it is element-wise, and the one sequential object -- the IBD chain, whose
# state at s is the Bernoulli(F) draw made at the last redraw site <= s -- is resumed exactly
# by looking back to that site.  Any (individual range) x (site range) slice is therefore bit
# for bit the slice of the whole data set.

_M64 = (1 << 64) - 1


def _i64(c):
    """A 64-bit constant as the int64 with the same bits."""
    c &= _M64
    return c - (1 << 64) if c >> 63 else c


_G1, _G2, _G3 = _i64(0x9E3779B97F4A7C15), _i64(0xD1B54A32D192ED03), _i64(0x8CB92BA72F3D8DD7)
_C1, _C2 = _i64(0xBF58476D1CE4E5B9), _i64(0x94D049BB133111EB)


def _lsr(x, k):
    return (x >> k) & ((1 << (64 - k)) - 1)


def _mix(x):
    """splitmix64's finaliser on int64 tensors (products wrap)."""
    x = (x ^ _lsr(x, 30)) * _C1
    x = (x ^ _lsr(x, 27)) * _C2
    return x ^ _lsr(x, 31)


def _mix_int(x):
    x &= _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


class IndexedSim:
    """The ngsF-HMMsim.R data model with every element addressed by its global indices.

    ``IndexedSim(I_tot, S_tot, device, ...)`` describes the whole data set; ``pos_dist(s0, s1)``
    and ``chunks(ind_range, site_range)`` produce any slice of it.  float parameters or "r"
    (uniform per individual / per site, ngsF-HMMsim.R:108-148).  ``true_params(ind_range)`` gives
    the (indF, alpha) the paths were drawn with."""

    LOOKBACK = 16384

    def __init__(self, n_ind, n_sites, device, *, freq=0.2, indF=0.5, alpha=0.01, depth=2.0,
                 error=0.01, seed=12345, n_chrom=1, missing_rate=0.0):
        import torch
        self.torch = torch
        self.I, self.S = int(n_ind), int(n_sites)
        self.device = device
        self.seed = int(seed)
        self.depth, self.error = float(depth), float(error)
        # cells without a read whatever the depth (a stream of its own: the other fields of the
        # data set do not depend on the rate)
        self.missing_rate = float(missing_rate)
        self.per_chr = -(-self.S // max(1, int(n_chrom)))
        self.n_chrom = int(n_chrom)
        self._freq, self._indF, self._alpha = freq, indF, alpha
        f64 = torch.float64
        self.p_read = torch.tensor([error, 0.5, 1.0 - error], device=device, dtype=f64)
        self.lp, self.lq = torch.log(self.p_read), torch.log1p(-self.p_read)
        # Poisson(depth) by inversion: cdf[k] = P(X <= k)
        kmax = int(depth + 12.0 * math.sqrt(depth) + 24)
        pk, cdf, acc = math.exp(-depth), [], 0.0
        for k in range(kmax):
            acc += pk
            cdf.append(min(acc, 1.0))
            pk *= depth / (k + 1)
        self.kmax = kmax
        self.pois_cdf = torch.tensor(cdf, device=device, dtype=f64)
        # Binomial(d, p_g) by inversion: T[d][g][k] = P(X <= k), 2.0 (never reached) for k >= d
        T = np.full((kmax + 1, 3, kmax), 2.0)
        for d in range(kmax + 1):
            for g, p in enumerate((error, 0.5, 1.0 - error)):
                acc = 0.0
                for k in range(d):
                    acc += math.comb(d, k) * p ** k * (1.0 - p) ** (d - k)
                    T[d, g, k] = acc
        self.binom_cdf = torch.tensor(T.reshape(-1), device=device, dtype=f64)

    # -- hashed uniforms ----------------------------------------------------------------------
    def _key(self, stream):
        return _i64(_mix_int(self.seed * 0x8CB92BA72F3D8DD7 + stream * 0x9E3779B97F4A7C15 + 1))

    def _u_site(self, stream, s_idx):
        """uniform [0, 1) per site: s_idx int64 tensor of global site indices."""
        h = _mix(_mix(s_idx * _G1 + self._key(stream)) + _G3)
        return _lsr(h, 11).to(self.torch.float64) * (1.0 / 9007199254740992.0)

    def _u(self, stream, i_idx, s_idx):
        """uniform [0, 1) per (individual, site): i_idx [I, 1], s_idx [1, n] global indices."""
        h = _mix(_mix(s_idx * _G1 + self._key(stream)) + i_idx * _G2)
        return _lsr(h, 11).to(self.torch.float64) * (1.0 / 9007199254740992.0)

    def _sites(self, s0, s1):
        return self.torch.arange(s0, s1, device=self.device, dtype=self.torch.int64)

    def _per_ind(self, what, stream, i_idx):
        if isinstance(what, str):
            return self._u_site(stream, i_idx)          # same hash family, another stream
        return self.torch.full(i_idx.shape, float(what), device=self.device, dtype=self.torch.float64)

    # -- per-site fields ------------------------------------------------------------------------
    def pos_dist(self, s0, s1):
        """Distances in Mb of the global sites [s0, s1) to their predecessors: the reader's
        (shared/read_data.cpp:165-218): d_0 = the first position, +inf at a chromosome change."""
        torch = self.torch
        s = self._sites(s0, s1)
        u1 = 1.0 - self._u_site(1, s)                    # (0, 1]
        u2 = self._u_site(2, s)
        z = torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * math.pi * u2)
        gaps = (1e5 + (1e5 / 3.0) * z).to(torch.int64).clamp_(min=1)
        d = gaps.to(torch.float64) / 1e6
        if self.n_chrom > 1:
            d[(s % self.per_chr == 0) & (s > 0)] = float("inf")
        return d

    def site_freq(self, s0, s1):
        s = self._sites(s0, s1)
        if isinstance(self._freq, str):
            return self._u_site(3, s)
        return self.torch.full((s1 - s0,), float(self._freq), device=self.device, dtype=self.torch.float64)

    # -- the IBD chain ----------------------------------------------------------------------------
    def _redraws(self, i_idx, F, A, sa, sb):
        torch = self.torch
        s = self._sites(sa, sb)[None, :]
        X = torch.exp(-A * self.pos_dist(sa, sb)[None, :])          # 0 at chromosome starts
        redraw = self._u(10, i_idx, s) >= X
        if sa == 0:
            redraw[:, 0] = True
        draws = (self._u(11, i_idx, s) < F).to(torch.int64)
        idx = torch.where(redraw, torch.arange(sb - sa, device=self.device)[None, :], -1)
        idx = torch.cummax(idx, dim=1).values
        return draws, idx

    def _state_before(self, i_idx, F, A, sa):
        """IBD state of every individual at global site sa - 1."""
        torch = self.torch
        if sa <= 0:
            return torch.zeros((i_idx.shape[0],), device=self.device, dtype=torch.int64)
        lo = max(0, sa - self.LOOKBACK)
        draws, idx = self._redraws(i_idx, F, A, lo, sa)
        last = idx[:, -1]
        st = torch.gather(draws, 1, last.clamp(min=0)[:, None])[:, 0]
        if lo > 0 and bool((last < 0).any()):
            st = torch.where(last >= 0, st, self._state_before(i_idx, F, A, lo))
        return st

    def true_params(self, ind_range=None):
        i0, i1 = ind_range if ind_range is not None else (0, self.I)
        i_idx = self.torch.arange(i0, i1, device=self.device, dtype=self.torch.int64)[:, None]
        return (self._per_ind(self._indF, 20, i_idx)[:, 0], self._per_ind(self._alpha, 21, i_idx)[:, 0])

    # -- slices -------------------------------------------------------------------------------
    def chunks(self, ind_range=None, site_range=None, chunk_sites=20000):
        """Generator of (site_begin - s0, gl [n][I_loc][3]) over the slice: normalised
        natural-log likelihoods, float64, file order."""
        torch = self.torch
        i0, i1 = ind_range if ind_range is not None else (0, self.I)
        s0, s1 = site_range if site_range is not None else (0, self.S)
        i_idx = torch.arange(i0, i1, device=self.device, dtype=torch.int64)[:, None]
        F = self._per_ind(self._indF, 20, i_idx)
        A = self._per_ind(self._alpha, 21, i_idx)
        state = self._state_before(i_idx, F, A, s0)
        for a in range(s0, s1, chunk_sites):
            b = min(s1, a + chunk_sites)
            s = self._sites(a, b)[None, :]
            draws, idx = self._redraws(i_idx, F, A, a, b)
            path = torch.where(idx >= 0, torch.gather(draws, 1, idx.clamp(min=0)), state[:, None])
            state = path[:, -1].clone()
            fr = self.site_freq(a, b)[None, :]
            h1 = (self._u(12, i_idx, s) < fr).to(torch.int64)
            h2 = (self._u(13, i_idx, s) < fr).to(torch.int64)
            h1 = torch.where(path == 1, h2, h1)
            geno = h1 + h2                                                   # [I][n]
            dep = torch.searchsorted(self.pois_cdf, self._u(14, i_idx, s), right=True)
            dep.clamp_(max=self.kmax)
            if self.missing_rate > 0:
                dep[self._u(16, i_idx, s) < self.missing_rate] = 0
            u = self._u(15, i_idx, s)
            base = (dep * 3 + geno) * self.kmax
            nA = torch.zeros_like(dep)
            for k in range(int(dep.max())):
                nA += (u >= self.binom_cdf[base + k]).to(torch.int64)
            del draws, idx, path, h1, h2, base, u
            nA = nA.to(torch.float64)
            dep = dep.to(torch.float64)
            ll = nA[..., None] * self.lp + (dep - nA)[..., None] * self.lq   # [I][n][3]
            del nA, dep, geno
            for _ in range(2):       # the simulator's normalisation, rounding, the reader's post_prob
                m = torch.maximum(torch.maximum(ll[..., 0], ll[..., 1]), ll[..., 2])
                e = torch.exp(ll[..., 0] - m) + torch.exp(ll[..., 1] - m) + torch.exp(ll[..., 2] - m)
                ll = ll - (m + torch.log(e))[..., None]
                if _ == 0:
                    ll = torch.round(ll, decimals=10)
            out = ll.permute(1, 0, 2).contiguous()
            del ll, m, e
            yield a - s0, out

    def gl(self, ind_range=None, site_range=None, chunk_sites=20000):
        """The slice as one device tensor [n_sites][n_ind][3]."""
        i0, i1 = ind_range if ind_range is not None else (0, self.I)
        s0, s1 = site_range if site_range is not None else (0, self.S)
        out = self.torch.empty((s1 - s0, i1 - i0, 3), device=self.device, dtype=self.torch.float64)
        for a, c in self.chunks(ind_range, site_range, chunk_sites):
            out[a:a + c.shape[0]] = c
            del c
        return out
