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
basic block of >= 60 instructions with its opcode histogram (the loop bodies).  A block ends at a
label OR behind a branch: the code a loop falls through to when it ends carries no label of its
own (until round 5 it was counted into the loop body above it -- est_maf's node loop read 208
instructions for its 139)."""
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
                ins = l.strip().split(";")[0].strip()
                cur[1].append(ins)
                if ins.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
                    cur = [cur[0] + "+", []]      # the fall-through part: a block of its own
                    blocks.append(cur)
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
        if "k_fast_estmaf<16" in title:
            estmaf_model(blocks)
        print()


def estmaf_model(blocks):
    """est_maf's parts by what only they contain (the blocks' order and sizes move with the
    compiler): set-up of the per-individual constants (the input selects), an exact pass = the
    lanes' sums with their packed pair reduction (v_permlane32_swap) + the serial recursion and the
    build decision (every block from the check to the node loop: an upper bound, the decision runs
    once or twice per site), the interpolant's check against an exact pass (three wave sums: the
    readlanes; once per site), a node evaluation (the loop that parks its sums in LDS), the
    addition of the parked sums behind it."""
    def vf(ins):
        c = Counter(x.split()[0] for x in ins)
        return (sum(v for k, v in c.items() if k.startswith("v_")), sum(v for k, v in c.items() if "f64" in k))
    def find(pred):
        return next((i for i, (_, ins) in enumerate(blocks) if pred(Counter(x.split()[0] for x in ins))), None)
    i_set = find(lambda c: c["v_cndmask_b32_e64"] + c["v_cndmask_b32_e32"] >= 30)
    i_pass = find(lambda c: c["v_permlane32_swap_b32_e32"] >= 2 and c["v_rcp_f64_e32"] >= 4)
    i_red = find(lambda c: c["v_readlane_b32"] >= 20)
    i_node = find(lambda c: c["ds_write_b128"] >= 1 and c["v_rcp_f64_e32"] == 4 and c["v_readlane_b32"] == 2)
    if None in (i_set, i_pass, i_red, i_node) or not (i_pass < i_red < i_node):
        print("   est_maf model: not recognised")
        return
    rec = [vf(ins) for _, ins in blocks[i_red + 1:i_node]]
    rec = (sum(v for v, _ in rec), sum(f for _, f in rec))
    tail = next((vf(ins) for _, ins in blocks[i_node + 1:i_node + 6]
                 if sum(1 for x in ins if x.startswith("ds_read_b128")) >= 8), (0, 0))
    parts = [("setup", vf(blocks[i_set][1])), ("pass", vf(blocks[i_pass][1])), ("check", vf(blocks[i_red][1])),
             ("recursion", rec), ("node", vf(blocks[i_node][1])), ("node_tail", tail)]
    print("   est_maf model (VALU/FP64 wave-instructions): " + ", ".join(f"{n} {v}/{f}" for n, (v, f) in parts))


if __name__ == "__main__":
    main()
