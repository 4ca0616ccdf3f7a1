#!/usr/bin/env python3
"""gpurun_out/parity_report.json (written by tests/conftest.py during `pytest tests -m gpu`) -> the text table kept under profiles/:
per test and quantity group the WORST achieved / bound ratio.   python3 tools/parity_report.py "<passed/skipped note>" > profiles/rNN_parity_report.txt"""
import json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rec = json.load(open(os.path.join(ROOT, "gpurun_out", "parity_report.json")))
worst = {}
for r in rec:
    key = (r["test"], r["quantity"].split("/")[0])
    ratio = r["achieved"] / max(r["bound"], 1e-30)
    if key not in worst or ratio > worst[key][0]:
        worst[key] = (ratio, r)
note = sys.argv[1] if len(sys.argv) > 1 else ""
print(f"# Achieved error next to its bound for every parity comparison of `python -m pytest tests -m gpu` ({note}):")
print(f"# {len(rec)} comparisons in {len({r['test'] for r in rec})} tests; per test and quantity group the WORST ratio achieved / bound is listed.")
print("# Written by tests/conftest.py (gpurun_out/parity_report.json) and summarised by tools/parity_report.py.  fp32 path: dtype0, bf16 path: dtype1.\n")
for (test, _), (ratio, r) in sorted(worst.items()):
    print(f"{test:86s} {r['quantity']:52s} {r['achieved']:10.3e}  bound {r['bound']:.1e}  ({100 * ratio:5.1f} %)")
top = sorted(worst.values(), key=lambda v: -v[0])[:8]
print("\n# tightest margins")
for ratio, r in top:
    print(f"# {100 * ratio:5.1f} %  {r['test']}  {r['quantity']}")
