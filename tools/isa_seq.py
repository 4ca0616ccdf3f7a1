#!/usr/bin/env python3
"""Instruction-class SEQUENCE of the MFMA-heaviest loop of one kernel (hipcc -S output, no GPU needed): how the compiler ordered matrix, VALU,
LDS and wait instructions -- M mfma, E transcendental, v other VALU, r LDS read, w LDS write, g global / buffer, | s_waitcnt, B s_barrier, s SALU.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -S --cuda-device-only mvlt_amd/csrc/attention.hip -o /tmp/attn.s
    python tools/isa_seq.py /tmp/attn.s attn_bwd_dma_kernelILi4ELi3E"""
import re
import sys
import textwrap


def main():
    lines = open(sys.argv[1]).read().split("\n")
    pat = sys.argv[2]
    start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l)][0]
    end = [i for i, l in enumerate(lines) if i > start and l.startswith(".Lfunc_end")][0]
    body = lines[start:end]
    labels = {m.group(1): j for j, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    best = None
    for j, l in enumerate(body):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < j:
            n = sum("v_mfma" in x for x in body[labels[m.group(1)]:j])
            if best is None or n > best[2]:
                best = (labels[m.group(1)], j, n)
    seq = []
    for l in body[best[0]:best[1]]:
        l = l.strip()
        if not l or l[0] in ".;/":
            continue
        op = l.split()[0]
        c = ("M" if op.startswith("v_mfma") else "E" if op.startswith(("v_exp", "v_rcp", "v_log", "v_rsq", "v_sqrt")) else
             "r" if op.startswith(("ds_read", "ds_load", "ds_bpermute")) else "w" if op.startswith("ds_") else "v" if op.startswith("v_") else
             "|" if op.startswith("s_waitcnt") else "B" if op.startswith("s_barrier") else "g" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else
             "s" if op.startswith("s_") else "?")
        seq.append(c)
    s = "".join(seq)
    print(f"{pat}: loop of {len(s)} instructions, {best[2]} MFMA")
    print("\n".join(textwrap.wrap(s, 160)))


if __name__ == "__main__":
    main()
