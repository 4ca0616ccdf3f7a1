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
for M, N, K in [(98304, 1280, 320), (49152, 2048, 512), (262144, 128, 1152)]:
    A = torch.randn(M, K, device=dev).to(bf); W = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf)
    out = torch.empty(M, N, device=dev, dtype=bf)
    for name, cc in [('normal', 0), ('no C store', 0x10000), ('no epilogue', 0x40000), ('1 k-tile only', 0x20000), ('1 k-tile, no store', 0x30000), ('1 k-tile, no epilogue', 0x60000)]:
        t = timeit(lambda: ops.gemm_nt(A, W, out, M, N, K, K, K, N, col_copies=cc))
        print('nt M=%d N=%d K=%d %-20s %.1f us' % (M, N, K, name, t * 1e3))
