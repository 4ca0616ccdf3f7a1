"""TN (wgrad) GEMM microbench: python tools/ubench_tn.py [reps]   (also the target of rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
from mvlt_amd._lib import conv3map
dev = torch.device('cuda:0')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bf = torch.bfloat16
def timeit(fn):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
SHAPES = [(49152, 2048, 512, None), (98304, 1280, 320, None), (262144, 192, 1728, 'conv'), (1081344, 64, 64, None), (294912, 128, 128, None),
          (49152, 512, 512, None), (98304, 320, 1280, None), (98304, 320, 320, None), (98304, 640, 320, None)]
for M, N1, N2, kind in SHAPES:
    A = torch.randn(M, N1, device=dev).to(bf)
    if kind == 'conv':
        C = N2 // 9
        Bm = torch.randn(M, C, device=dev).to(bf)
        kw = dict(b_map=conv3map(32, 32, 1024, C)); ldb = C
    else:
        Bm = torch.randn(M, N2, device=dev).to(bf); kw = {}; ldb = N2
    out = torch.zeros(N1, N2, device=dev)
    cs = torch.zeros(N1, device=dev)
    t = timeit(lambda: ops.gemm_tn(A, Bm, out, M, N1, N2, N1, ldb, N2, colsum=None if kind else cs, **kw))
    print('tn M=%d N1=%d N2=%d %s: %.1f us  %.0f TF/s  %.0f GB/s(alg)' % (M, N1, N2, kind or '', t * 1e3, 2.0 * M * N1 * N2 / t / 1e9,
          (A.numel() + Bm.numel()) * 2 / t / 1e6))
    del A, Bm
