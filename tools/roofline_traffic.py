#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE counter CSVs of tools/roofline_launch.py into per-launch HBM traffic.

    python tools/roofline_traffic.py gpurun_out/pmc_rd gpurun_out/pmc_wr profiles/r01_roofline_traffic.json

Units and corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB, collected in SEPARATE --pmc passes
(FETCH_SIZE takes 3 of the 4 TCC slots).  On gfx950 FETCH_SIZE reports exactly half of the bytes of wide (16 B/lane)
coalesced streaming reads, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is taken as reported.  Both
corrections are re-checked in the same run against a kernel of known traffic: torch's fp32->bf16 cast of the conv input
(reads 4 B, writes 2 B per element) that tools/roofline_launch.py happens to run while building its operands.
Values are averaged over the 5 dispatches of each kernel.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd.build import source_hash  # noqa: E402

CASE = {"gemm_tn_p8_kernel<3, 3, 2": "tn_s3dw2", "gemm_tn_dma_kernel<128, 128, 3, 2, false>": "tn_s3dw2", "mlp_wgrad2_kernel<64": "mlp_dw64", "mlp_wgrad_kernel<64": "mlp_dw64", "conv3_nt_kernel<32, 192": "conv192", "gemm_nt_dma_kernel<192": "conv192", "gemm_nt_dma_kernel<128": "conv192", "gemm_nt_dma_kernel<64": "proj64", "gemm_nt_kernelIDF16bLi128": "conv192",
        "gemm_nt_kernelIDF16bLi64": "proj64", "bfloat16_copy_kernel": "calib_cast"}


def load(d, counter):
    out = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for pat, case in CASE.items():
                if pat in r["Kernel_Name"]:
                    if case == "calib_cast" and int(r["Grid_Size"]) != 6291456:      # the 262144x192 conv input only
                        continue
                    out[case].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}, {k: len(v) for k, v in out.items()}


if __name__ == "__main__":
    rd, wr, dst = sys.argv[1:4]
    (R, nR), (W, nW) = load(rd, "FETCH_SIZE"), load(wr, "WRITE_SIZE")
    res = {}
    for case in R:
        rb, wb = 2 * R[case] * 1024, W[case] * 1024
        res[case] = dict(dispatches=nR[case], fetch_size_kib=R[case], write_size_kib=W[case], read_bytes=rb, write_bytes=wb,
                         hbm_bytes=rb + wb)
    n = 262144 * 192
    res["calib_cast"]["expected_read_bytes"] = 4 * n
    res["calib_cast"]["expected_write_bytes"] = 2 * n
    res["_source_hash"] = source_hash()          # the kernel sources these counters were collected for (bench.py checks it)
    json.dump(res, open(dst, "w"), indent=1)
    print(json.dumps(res, indent=1))
