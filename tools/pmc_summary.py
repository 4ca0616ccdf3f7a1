#!/usr/bin/env python3
"""Per-kernel mean of every counter in a rocprofv3 counter_collection.csv: python tools/pmc_summary.py <dir> [name filter]"""
import csv, glob, sys
from collections import defaultdict
d = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if flt not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(")[0][-60:] + " grid=" + r["Grid_Size"]
        d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in d:
    print(k, " n=%d  mean %.1f us" % (len(dur[k]) // max(1, len(d[k])), sum(dur[k]) / len(dur[k]) / 1e3))
    print("    " + "  ".join("%s=%.3g" % (c, sum(v) / len(v)) for c, v in sorted(d[k].items())))
