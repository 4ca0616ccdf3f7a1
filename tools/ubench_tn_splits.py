import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16
def timeit(fn, reps=20):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for M, N1, N2, sp in [(49152, 2048, 512, (0, 8, 16)), (98304, 1280, 320, (0, 16)), (1081344, 64, 64, (0, 256, 512)),
                      (294912, 128, 128, (0,256)), (49152, 512, 512, (0,32))]:
    A = torch.randn(M, N1, device=dev).to(bf); Bm = torch.randn(M, N2, device=dev).to(bf)
    out = torch.zeros(N1, N2, device=dev)
    for s in sp:
        t = timeit(lambda: ops.gemm_tn(A, Bm, out, M, N1, N2, N1, N2, N2, splits=s))
        cs = torch.zeros(N1, device=dev)
        t2 = timeit(lambda: ops.gemm_tn(A, Bm, out, M, N1, N2, N1, N2, N2, splits=s, colsum=cs))
        print('tn M=%d N1=%d N2=%d splits=%d: %.1f us  %.0f TF/s   with colsum %.1f us' % (M, N1, N2, s, t * 1e3, 2.0 * M * N1 * N2 / t / 1e9, t2 * 1e3))
