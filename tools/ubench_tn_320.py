"""The stage-3 fc weight-gradient shapes (1280 x 320 and 320 x 1280 over M rows) with partial tiles, as the step launches them: python tools/ubench_tn_320.py [M] [reps]
(MVLT_TN_P8_320=0: the 128-wide kernel; also the target of rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
from mvlt_amd._lib import last_kernel
dev = torch.device('cuda:0'); bf = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 98304
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
scr = torch.empty(256 * 8 * 65536, dtype=bf, device=dev)
for N1, N2 in ((1280, 320), (320, 1280)):
    A, B = torch.randn(M, N1, device=dev).to(bf), torch.randn(M, N2, device=dev).to(bf)
    out, cs = torch.zeros(N1, N2, device=dev), torch.zeros(N1, device=dev)
    def fn():
        ops.gemm_tn(A, B, out, M, N1, N2, N1, N2, N2, colsum=cs, partials=scr, defer_fold=True)
        k = last_kernel()
        ops.tn_fold_flush(scr)
        return k
    for _ in range(2): name = fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps
    print('tn M=%d %d x %d: %.1f us incl. fold  %.0f TF/s  [%s]' % (M, N1, N2, t * 1e3, 2.0 * M * N1 * N2 / t / 1e9, name))
