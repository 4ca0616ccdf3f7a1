#!/usr/bin/env python3
"""Launches of ONE steady-state training step from a rocprofv3 kernel trace of bench.py (tools/profile_bench.sh): the kernels between two
consecutive fused-AdamW launches, counted by family.  (Averages over the whole bench process -- round 3's "512 launches incl. 49.5 copyBuffer
per step" -- include model construction, the synthetic batch and the warm-up's one-off copies.)
    python tools/step_launches.py gpurun_out/prof_r04/trace_kernel_trace.csv > profiles/r04_step_launches.txt"""
import collections
import csv
import re
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
lo, hi = ends[-4] + 1, ends[-3] + 1
step = rows[lo:hi]


def fam(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    n = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", n)
    m = re.match(r"([A-Za-z_0-9:]+)", n)
    return (m.group(1) if m else n)[:44]


cnt, dur = collections.Counter(), collections.Counter()
for r in step:
    f = fam(r["Kernel_Name"])
    cnt[f] += 1
    dur[f] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
busy = sum(dur.values())
span = int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])
aten = sum(c for f, c in cnt.items() if f.startswith("at::"))
print(f"one steady-state step: {len(step)} launches, {span / 1e6:.3f} ms from the first kernel's start to the optimizer's end, kernels busy {busy / 1e6:.3f} ms "
      f"({100.0 * busy / span:.1f} %); ATen kernels {aten}, __amd_rocclr_copyBuffer {cnt.get('__amd_rocclr_copyBuffer', 0)}, fills {cnt.get('__amd_rocclr_fillBufferAligned', 0)}")
print(f"{'launches':>8s} {'us/step':>9s}  kernel family")
for f, c in sorted(cnt.items(), key=lambda kv: -dur[kv[0]]):
    print(f"{c:8d} {dur[f] / 1e3:9.1f}  {f}")
