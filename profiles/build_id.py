#!/usr/bin/env python3
"""Identity of the build a committed profile belongs to: sha256 over the sources of the library
(ngsf-hmm_amd/csrc/**: .hip .hpp .h .cpp Makefile, and include/nghmm.h, include/nghmm_debug.h), path and content, in
sorted order -- 16 hex digits.  profiles/collect.sh stores it in rNN_pmc_summary.json,
tools/isa_report.py in rNN_isa_summary.txt, bench.py --write_check in check_n1.json; bench.py
compares it with the sources it runs from and drops replayed counters (HBM traffic, instruction
counts, the one-GPU check) that belong to another build instead of quoting them.

  python profiles/build_id.py        prints the id of this tree"""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXT = (".hip", ".hpp", ".h", ".cpp")


def build_id(root=ROOT):
    files = [os.path.join(root, "include", "nghmm.h"), os.path.join(root, "include", "nghmm_debug.h")]
    csrc = os.path.join(root, "ngsf-hmm_amd", "csrc")
    for d, _, names in os.walk(csrc):
        for n in names:
            if n.endswith(EXT) or n == "Makefile":
                files.append(os.path.join(d, n))
    h = hashlib.sha256()
    for f in sorted(files):
        h.update(os.path.relpath(f, root).encode())
        h.update(b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(build_id())
