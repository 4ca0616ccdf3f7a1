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
M, N, K = 1490, 768, 30528
A = torch.randn(M, K, device=dev).to(bf); W = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf)
out = torch.zeros(M, N, device=dev, dtype=torch.float32)
for S in (0, 2, 4, 8, 16, 32):
    t = timeit(lambda: ops.gemm_nt(A, W, out, M, N, K, K, K, N, split_k=S))
    print('split_k=%d  %.1f us' % (S, t * 1e3))
