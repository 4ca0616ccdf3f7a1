"""A handful of NT GEMM shapes of the step, timed alone: python tools/ubench_nt_p8.py   (A/B by MVLT_NT_P8 / MVLT_HIP_LIB in separate processes)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for M, N, K, res, odt in ((98304, 320, 320, True, torch.float32), (98304, 320, 320, False, bf), (49152, 512, 512, True, torch.float32), (49152, 512, 512, False, bf),
                          (49152, 1024, 512, False, bf), (49152, 512, 1024, True, bf), (98304, 320, 1280, True, torch.float32), (49152, 512, 2048, True, torch.float32)):
    A = torch.randn(M, K, device=dev).to(bf); W = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf); b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=odt); R = torch.randn(M, N, device=dev).to(odt) if res else None
    t = timeit(lambda: ops.gemm_nt(A, W, out, M, N, K, K, K, N, bias=b, R=R))
    print('%6d x %4d x %4d %s %-8s %7.1f us %6.0f TF' % (M, N, K, '+R' if res else '  ', str(odt)[6:], t, 2.0 * M * N * K / t / 1e6), flush=True)
