"""The 192 -> 192 conv3x3 as the gathered NT GEMM (roofline launch) with and without the statistics epilogue."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
from mvlt_amd._lib import conv3map
dev = torch.device('cuda:0'); bf = torch.bfloat16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
def timeit(fn):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
B, side = 256, 32
for C, N in [(192, 192), (128, 128), (64, 64)]:
    M = B * side * side
    x = torch.randn(M, C, device=dev).to(bf); W = (torch.randn(N, 9 * C, device=dev) * 0.02).to(bf)
    out = torch.empty(M, N, device=dev); st = torch.zeros(2, 16, N, device=dev)
    amap = conv3map(side, side, side * side, C)
    t1 = timeit(lambda: ops.gemm_nt(x, W, out, M, N, 9 * C, C, 9 * C, N, a_map=amap))
    t2 = timeit(lambda: ops.gemm_nt(x, W, out, M, N, 9 * C, C, 9 * C, N, a_map=amap, col_sum=st[0], col_sumsq=st[1], col_copies=16))
    fl = 2.0 * M * N * 9 * C
    print('conv %d->%d plain %.1f us %.0f TF/s | stats %.1f us %.0f TF/s' % (C, N, t1 * 1e3, fl / t1 / 1e9, t2 * 1e3, fl / t2 / 1e9))
