#!/usr/bin/env python3
"""Per-kernel SQ instruction counters over whole training steps (tools/step_sq.sh): sums per kernel name and per step of every counter of the pass.
    python3 tools/step_sq.py <dir of the pass>  -> CSV on stdout: kernel, calls_per_step, us_per_step, <counter>_per_step ..."""
import csv, glob, os, re, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd.build import source_hash  # noqa: E402

val = defaultdict(lambda: defaultdict(float))
dur, calls, seen = defaultdict(float), defaultdict(int), set()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"^void ", "", n)[:70]
        val[n][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], n)
        if key not in seen:
            seen.add(key)
            dur[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
            calls[n] += 1
steps = max([c for n, c in calls.items() if n.startswith("adamw_kernel")] or [1])
counters = sorted({c for v in val.values() for c in v})
w = csv.writer(sys.stdout)
w.writerow([f"kernel [source_hash={source_hash()} steps={steps}]", "calls_per_step", "us_per_step"] + counters)
for n in sorted(val, key=lambda k: -dur[k]):
    w.writerow([n, round(calls[n] / steps, 2), round(dur[n] / steps, 1)] + [round(val[n].get(c, 0.0) / steps, 1) for c in counters])
