#!/usr/bin/env python3
"""Instruction mix of the hottest loop of every kernel in a HIP source (hipcc -S, no GPU needed):
    python tools/isa_mix.py mvlt_amd/csrc/mlp.hip [name filter]
Per kernel: VGPR / AGPR / LDS / occupancy from the metadata, and for the largest loop (label .. backward branch) the number of
MFMA, transcendental (quarter rate), packed, accvgpr-move, other VALU, LDS, VMEM, SALU and s_waitcnt instructions.  The VALU
issue estimate counts 4 cycles per VALU instruction and 16 per transcendental and per 16x16x32 bf16 MFMA."""
import re
import subprocess
import sys
from collections import Counter


def main():
    src, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    import os
    extra = os.environ.get("MVLT_ISA_FLAGS", "").split() + (["-fno-slp-vectorize"] if src.endswith("mlp.hip") else [])    # the build's own flags
    asm = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", *extra, "-S", "--cuda-device-only", src, "-o", "-"],
                         capture_output=True, text=True).stdout
    lines = asm.split("\n")
    starts = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"^(_Z\w+):\s*(;.*)?$", l)] if m]
    for n, (i0, name) in enumerate(starts):
        i1 = starts[n + 1][0] if n + 1 < len(starts) else len(lines)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if flt not in dem:
            continue
        body = lines[i0:i1]
        meta = {k: v for l in body for k, v in re.findall(r"; (NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize): (\d+)", l)}
        labels = {m.group(1): j for j, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        best = None
        for j, l in enumerate(body):
            m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < j:
                if best is None or j - labels[m.group(1)] > best[1] - best[0]:
                    best = (labels[m.group(1)], j)
        print(dem[:110])
        print("   ", meta)
        if best is None:
            continue
        c = Counter()
        for l in body[best[0]:best[1]]:
            l = l.strip()
            if not l or l[0] in ".;/":
                continue
            op = l.split()[0]
            if op.startswith("v_mfma"):
                c["mfma"] += 1
            elif op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")):
                c["trans"] += 1
            elif op.startswith("v_accvgpr"):
                c["accmov"] += 1
            elif op.startswith("v_pk_"):
                c["pk"] += 1
            elif op.startswith("v_"):
                c["valu"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                c["vmem"] += 1
            elif op.startswith("s_waitcnt"):
                c["waitcnt"] += 1
            elif op.startswith("s_barrier"):
                c["barrier"] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
        valu_cyc = 4 * (c["pk"] + c["valu"] + c["accmov"]) + 16 * c["trans"]
        print("    loop:", dict(c), f" VALU issue ~{valu_cyc} cyc, MFMA ~{16 * c['mfma']} cyc (16x16x32)")


if __name__ == "__main__":
    main()
