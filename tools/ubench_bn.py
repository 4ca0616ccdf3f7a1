"""Micro-benchmark of the MIM decoder's BatchNorm passes (csrc/mim.hip) at the step's shapes: z fp32 against fp16.
    gpurun -- python tools/ubench_bn.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvlt_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for M, C in ((262144, 64), (262144, 128), (262144, 192), (65536, 128)):
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    mean, rstd, gamma, beta = (torch.randn(C, device=dev) for _ in range(4))
    rstd = rstd.abs() + 0.5
    dy = torch.randn(M, C, device=dev).to(torch.bfloat16)
    red = torch.zeros(2, C, device=dev)
    y16 = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
    y32 = torch.empty(M, C, device=dev)
    dz = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
    for zt in (torch.float32, torch.float16):
        z = torch.randn(M, C, device=dev).to(zt)

        def wrap(f):
            def g():
                flush.zero_()          # push the operands out of the Infinity Cache (timed too: subtract the flush alone)
                f()
            return g
        t0 = timeit(wrap(lambda: None))
        tn16 = timeit(wrap(lambda: ops.bn_norm(z, C, mean, rstd, gamma, beta, M, C, y16=y16, ld16=C))) - t0
        tn32 = timeit(wrap(lambda: ops.bn_norm(z, C, mean, rstd, gamma, beta, M, C, y32=y32, ld32=C))) - t0
        tr = timeit(wrap(lambda: ops.bn_bwd_reduce(dy, C, z, C, mean, rstd, M, C, red[0], red[1]))) - t0
        ta = timeit(wrap(lambda: ops.bn_bwd_apply(dy, C, z, C, mean, rstd, gamma, red[0], red[1], M, C, dz, C))) - t0
        # warm (operands resident in the cache hierarchy where they fit): what the step sees right after the producer
        wn16 = timeit(lambda: ops.bn_norm(z, C, mean, rstd, gamma, beta, M, C, y16=y16, ld16=C))
        wr = timeit(lambda: ops.bn_bwd_reduce(dy, C, z, C, mean, rstd, M, C, red[0], red[1]))
        wa = timeit(lambda: ops.bn_bwd_apply(dy, C, z, C, mean, rstd, gamma, red[0], red[1], M, C, dz, C))
        print(f"M={M} C={C} z={str(zt)[6:]:8s} cold: norm->bf16 {tn16:6.1f} norm->f32 {tn32:6.1f} reduce {tr:6.1f} apply {ta:6.1f} us | warm: norm {wn16:6.1f} reduce {wr:6.1f} apply {wa:6.1f}")
