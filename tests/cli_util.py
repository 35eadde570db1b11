"""Helpers for the command-line tests: write reference-format input files, and build
the expected .indF / .ibd / .geno bytes (EM.cpp:293-380) from oracle results."""
import gzip
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINARY = os.path.join(ROOT, "ngsf-hmm_amd", "ngsF-HMM")


def write_inputs(tmp, d, gl_raw):
    """gl_raw: [S][I][3] log GLs as the simulator emits them (not yet normalised)."""
    S, I = gl_raw.shape[:2]
    paths = {}
    paths["glf_gz"] = os.path.join(tmp, "sim.glf.gz")
    with gzip.open(paths["glf_gz"], "wt") as fh:
        for s in range(S):
            fh.write("\t".join(repr(float(v)) for v in gl_raw[s].reshape(-1)) + "\n")
    # BEAGLE-style: marker + two allele columns, then 3 normal-space likelihoods per
    # individual (BASELINE.json configs[2] input; the reader takes the LAST 3*I columns,
    # shared/read_data.cpp:84, and --lkl without --loglkl takes their log, :89).  The header
    # has no numeric token, so split() returns 0 fields and the line is skipped
    # (gen_func.cpp:390-417, read_data.cpp:64).
    paths["beagle_gz"] = os.path.join(tmp, "sim.beagle.gz")
    with gzip.open(paths["beagle_gz"], "wt") as fh:
        fh.write("marker\tallele1\tallele2\t" +
                 "\t".join(f"Ind{i}" for i in range(I) for _ in range(3)) + "\n")
        for s in range(S):
            fh.write(f"chr{int(d.chrom[s])}_{int(d.pos[s])}\t0\t1\t" +
                     "\t".join(repr(float(v)) for v in np.exp(gl_raw[s]).reshape(-1)) + "\n")
    paths["glf_bin"] = os.path.join(tmp, "sim.glf")
    gl_raw.astype("<f8").tofile(paths["glf_bin"])
    paths["geno_gz"] = os.path.join(tmp, "sim.geno.gz")
    with gzip.open(paths["geno_gz"], "wt") as fh:
        for s in range(S):
            fh.write("\t".join(str(int(v)) for v in d.geno[s]) + "\n")
    paths["pos_gz"] = os.path.join(tmp, "sim.pos.gz")
    with gzip.open(paths["pos_gz"], "wt") as fh:
        for s in range(S):
            fh.write(f"chr{int(d.chrom[s])}\t{int(d.pos[s])}\t0.2\n")
    return paths


def raw_called_genotypes(geno):
    """What the reader makes of a called-genotype file (shared/read_data.cpp:21,91-97):
    [S][I][3] with log(1) at the called genotype, -1e15 elsewhere, log(1/3) x 3 if missing."""
    S, I = geno.shape
    raw = np.full((S, I, 3), -1e15)
    for g in range(3):
        raw[..., g][geno == g] = 0.0
    raw[geno < 0] = np.log(1.0 / 3.0)
    return raw


def expected_files(tot_lkl, indF, alpha, freq, ind_lkl, path, marg, geno_post):
    """Bytes of PREFIX.indF / .ibd / .geno as EM.cpp:293-380 prints them."""
    I, S = path.shape
    lines = ["%.10f\n" % tot_lkl]
    for i in range(I):
        if indF[i] < 1e-5:
            lines.append("%.5f\tNA\n" % 0.0)
        elif indF[i] > 1 - 1e-5:
            lines.append("%.5f\tNA\n" % 1.0)
        else:
            lines.append("%.5f\t%f\n" % (indF[i], alpha[i]))
    lines += ["%f\n" % f for f in freq]
    f_indF = "".join(lines).encode()
    out = ["//\t" + "\t".join("%.10f" % v for v in ind_lkl) + "\n"]
    for i in range(I):
        out.append("".join(chr(48 + int(v)) for v in path[i]) + "\n")
    for i in range(I):
        out.append("\t".join("%f" % v for v in marg[i]) + "\n")
    f_ibd = "".join(out).encode()
    f_geno = np.ascontiguousarray(geno_post, dtype="<f8").tobytes()
    return f_indF, f_ibd, f_geno


def run_cli(args, check=True, env=None):
    return subprocess.run([BINARY] + [str(a) for a in args], capture_output=True, text=True,
                          check=check, env=env)


def write_bgzf(path, data, block=0xff00):
    """`data` (bytes) as a BGZF file: gzip members of at most `block` bytes of data, each with
    the 'BC' extra field holding the member's size - 1, and the empty member that ends the file
    (the SAM specification's section on BGZF; what bgzip and ANGSD write)."""
    import struct
    import zlib
    with open(path, "wb") as fh:
        for o in list(range(0, len(data), block)) + [None]:
            piece = b"" if o is None else data[o:o + block]
            c = zlib.compressobj(6, zlib.DEFLATED, -15)
            z = c.compress(piece) + c.flush()
            bsize = 18 + len(z) + 8
            assert bsize <= 65536
            fh.write(bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0]) + b"BC" +
                     struct.pack("<HH", 2, bsize - 1) + z +
                     struct.pack("<II", zlib.crc32(piece) & 0xffffffff, len(piece)))
