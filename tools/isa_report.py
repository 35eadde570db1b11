#!/usr/bin/env python3
"""Per-loop instruction counts of the two FP64-bound kernels from the device assembly:

  for f in walks estep estmaf; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only \
          ngsf-hmm_amd/csrc/kernels_fast_$f.hip -o /tmp/kf_$f.s
  done
  cat /tmp/kf_walks.s /tmp/kf_estep.s /tmp/kf_estmaf.s > /tmp/kf.s
  python tools/isa_report.py /tmp/kf.s > profiles/r05_isa_summary.txt
  (tools/isa_summary.sh does all of it)

For each kernel: registers / scratch / occupancy as the assembler reports them, and every
basic block of >= 60 instructions with its opcode histogram (the loop bodies)."""
import re
import sys
from collections import Counter

KERNELS = {
    "later objective rounds: k_fast_lkl_fd<2, 2, true, false, SRC_PLAIN, 2> (8 sites per loop body)":
        "_ZN5nghmm12_GLOBAL__N_113k_fast_lkl_fdILi2ELi2ELb1ELb0ELi0ELi2ELb0EEE",
    "fresh forward walk: k_fast_lkl_fd<2, 2, true, true, SRC_FRESH, 2>":
        "_ZN5nghmm12_GLOBAL__N_113k_fast_lkl_fdILi2ELi2ELb1ELb1ELi1ELi2ELb0EEE",
    "est_maf: k_fast_estmaf<16, 64, true> (1000 individuals per site: 16 per lane)":
        "_ZN5nghmm12_GLOBAL__N_113k_fast_estmafILi16ELi64ELb1EEE",
    "backward sweep: k_fast_bwd_recompute8 (4 waves = 8 individuals x 32 lane-chunks, LDS staging)":
        "_ZN5nghmm12_GLOBAL__N_121k_fast_bwd_recompute8",
    "est_maf interpolated passes: k_fast_estmaf_interp": "_ZN5nghmm12_GLOBAL__N_120k_fast_estmaf_interp",
    "est_maf, called genotypes: k_fast_estmaf_called_sums<true> (one sweep over codes and posteriors)":
        "_ZN5nghmm12_GLOBAL__N_125k_fast_estmaf_called_sumsILb1EEE",
}


def main():
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles"))
    from build_id import build_id
    print("# build_id:", build_id(), "(profiles/build_id.py: the sources this assembly was made from)")
    lines = open(sys.argv[1]).read().split("\n")
    for title, sym in KERNELS.items():
        start = [i for i, l in enumerate(lines) if l.startswith(sym)][0]
        end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
        blocks, cur = [], None
        for l in lines[start:end]:
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                cur = [m.group(1), []]
                blocks.append(cur)
            elif cur is not None and l.startswith("\t") and not l.strip().startswith((".", ";")):
                cur[1].append(l.strip().split(";")[0].strip())
        print("==", title)
        for l in lines[end:end + 60]:
            if any(t in l for t in ("NumVgprs:", "ScratchSize:", "Occupancy:", "NumSgprs:")):
                print("  ", l.strip("; ").strip())
        print("   total instructions:", sum(len(b[1]) for b in blocks), "in", len(blocks), "basic blocks")
        for name, ins in blocks:
            if len(ins) < 60:
                continue
            c = Counter(x.split()[0] for x in ins)
            f64 = sum(v for k, v in c.items() if "f64" in k)
            valu = sum(v for k, v in c.items() if k.startswith("v_"))
            vmem = sum(v for k, v in c.items() if k.startswith(("global_", "buffer_", "flat_")))
            lds = sum(v for k, v in c.items() if k.startswith("ds_"))
            salu = sum(v for k, v in c.items() if k.startswith("s_"))
            print(f"   block {name}: {len(ins)} instructions = {valu} VALU ({f64} FP64) + {vmem} VMEM + "
                  f"{lds} LDS + {salu} scalar")
            print("      ", ", ".join(f"{k} {v}" for k, v in c.most_common(12)))
        print()


if __name__ == "__main__":
    main()
