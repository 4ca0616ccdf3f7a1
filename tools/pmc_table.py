#!/usr/bin/env python3
"""Per-kernel means of every counter collected by tools/pmc_collect.sh, plus derived ratios, as CSV:
    python tools/pmc_table.py gpurun_out/pmc_r02_ > profiles/r02_mfma_counters.csv
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave, SQ_BUSY_CYCLES per SE-ish unit, SQ_VALU_MFMA_BUSY_CYCLES
cycles per SIMD (MI355X_MICROARCH.md): the derived columns are ratios within one kernel, the raw values stay beside them."""
import csv
import glob
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([\w:]+(?:<[^()]*?>)?)\(", name)
    return (m.group(1) if m else name)[:90]


def main():
    vals = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for f in sorted(glob.glob(sys.argv[1] + "*/**/*counter_collection.csv", recursive=True)):
        seen = set()
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if not ("mvlt" in n or "_GLOBAL__N_1" in n or "anonymous namespace" in n) or "at::" in n:
                continue
            k = short(n) + " grid=" + r["Grid_Size"]
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (f, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    counters = sorted({c for v in vals.values() for c in v})
    derived = ["us_mean", "mfma_busy_frac_of_simd_cycles", "wait_any_frac", "wait_inst_any_frac", "active_inst_frac", "valu_active_frac", "lds_conflict_frac"]
    w = csv.writer(sys.stdout)
    w.writerow(["kernel"] + derived + counters)
    for k in sorted(vals):
        m = {c: sum(v) / len(v) for c, v in vals[k].items()}
        us = sum(dur[k]) / len(dur[k])
        wc = m.get("SQ_WAVE_CYCLES", 0)
        gui = m.get("GRBM_GUI_ACTIVE", 0)
        d = [us,
             m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui / 8.0 * 256 * 4) if gui else "",     # busy cycles summed over the 1024 SIMDs / (shader cycles x 1024);
                                                                                               # GRBM_GUI_ACTIVE arrives summed over the 8 XCDs (was 8x too small)
             m.get("SQ_WAIT_ANY", 0) / wc if wc else "", m.get("SQ_WAIT_INST_ANY", 0) / wc if wc else "",
             m.get("SQ_ACTIVE_INST_ANY", 0) / wc if wc else "", m.get("SQ_ACTIVE_INST_VALU", 0) / wc if wc else "",
             m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"] if m.get("SQ_LDS_IDX_ACTIVE") else ""]
        w.writerow([k] + [("%.4g" % v if v != "" else "") for v in d] + ["%.6g" % m[c] if c in m else "" for c in counters])


if __name__ == "__main__":
    main()
