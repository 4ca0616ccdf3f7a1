#!/usr/bin/env python3
"""HBM bytes per kernel over whole training steps: two rocprofv3 passes (FETCH_SIZE, WRITE_SIZE, each with --kernel-trace only) over
`bench.py --steps S --warmup 0 --no-cpu-baseline`, joined per kernel name.
    python3 tools/step_traffic.py <dir of FETCH_SIZE pass> <dir of WRITE_SIZE pass> [steps]   -> table on stdout (steps = AdamW launches seen)
read bytes = 2 x FETCH_SIZE KiB (gfx950 half-count correction, MI355X_MICROARCH.md), written = WRITE_SIZE KiB."""
import csv, glob, os, re, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd.build import source_hash  # noqa: E402


def load(d, counter):
    val, dur, calls = defaultdict(float), defaultdict(float), defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            n = re.sub(r"^void ", "", n)[:70]
            val[n] += float(r["Counter_Value"])
            dur[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
            calls[n] += 1
    return val, dur, calls


if __name__ == "__main__":
    rd, wr = sys.argv[1], sys.argv[2]
    R, durR, calls = load(rd, "FETCH_SIZE")
    W, durW, _ = load(wr, "WRITE_SIZE")
    # steps actually executed (bench.py adds its blocks-only timing iterations to --steps): one AdamW launch per step
    steps = max([c for n, c in calls.items() if n.startswith("adamw_kernel")] or [int(sys.argv[3])])
    rows = []
    for n in R:
        rb, wb = 2 * R[n] * 1024 / steps, W.get(n, 0.0) * 1024 / steps
        us = 0.5 * (durR[n] + durW.get(n, durR[n])) / steps
        rows.append((rb + wb, rb, wb, us, calls[n] / steps, n))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"total HBM bytes per step (all launches of the bench process, setup included, over the {steps} steps it ran): {tot / 1e9:.2f} GB   [source_hash={source_hash()}]")
    print(f"{'MB/step':>9} {'read':>8} {'written':>8} {'us/step':>9} {'TB/s':>6} {'calls':>6}  kernel")
    for t, rb, wb, us, c, n in rows:
        print(f"{t / 1e6:9.1f} {rb / 1e6:8.1f} {wb / 1e6:8.1f} {us:9.1f} {t / us / 1e6 if us else 0:6.2f} {c:6.1f}  {n}")
