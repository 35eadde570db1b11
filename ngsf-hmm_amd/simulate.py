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
