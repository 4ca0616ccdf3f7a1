#!/usr/bin/env python3
"""Launch only the two roofline kernels of bench.py (5x each) so that a rocprofv3 --pmc pass sees nothing else.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_rd -o rd -- python3 tools/roofline_launch.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_wr -o wr -- python3 tools/roofline_launch.py
then tools/roofline_traffic.py turns the two counter CSVs into profiles/<round>_roofline_traffic.json.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    for name, fn, _ in bench.roofline_cases(int(os.environ.get("B", "256")), dev):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
