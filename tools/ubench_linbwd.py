"""weight + input gradient of the C x C Linears of stages 1-2: two launches against the fused pass over dY"""
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
for M, C in ((1081344, 64), (294912, 128)):
    dY, X = torch.randn(M, C, device=dev).to(bf), torch.randn(M, C, device=dev).to(bf)
    WT = (torch.randn(C, C, device=dev) * C ** -0.5).to(bf)
    dW, db, dX = torch.zeros(C, C, device=dev), torch.zeros(C, device=dev), torch.empty(M, C, device=dev, dtype=bf)
    def two():
        ops.gemm_tn(dY, X, dW, M, C, C, C, C, C, colsum=db)
        ops.gemm_nt(dY, WT, dX, M, C, C, C, C, C)
    t2 = timeit(two)
    t1 = timeit(lambda: ops.gemm_tn(dY, X, dW, M, C, C, C, C, C, colsum=db, dgrad=(WT, dX)))
    print('M=%d C=%d: two launches %.1f us | fused %.1f us (%.2f TB/s over dY + X + dX)' % (M, C, t2, t1, 3 * M * C * 2 / t1 / 1e6), flush=True)
    for sp in (128, 192, 256, 384, 512, 768):
        t = timeit(lambda: ops.gemm_tn(dY, X, dW, M, C, C, C, C, C, colsum=db, dgrad=(WT, dX), splits=sp))
        print('    splits %4d: %.1f us' % (sp, t), flush=True)
