"""The MLM decoder's weight gradient alone: dW[30522, 768] += dlogits[rows, 30522]^T hidden[rows, 768] (+ bias gradient), rows = the ~1500 selected
tokens of a batch -- ONE m-split, 1434 output tiles.  A/B of the plain accumulate against the fp32 atomics: MVLT_HIP_LIB=ab/libmvlt_tnatomic.so."""
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
for M in (1490, 3000):
    V, K = 30522, 768
    ld = (V + 7) // 8 * 8
    dl = torch.zeros(M, ld, device=dev, dtype=bf); dl[:, :V] = (torch.randn(M, V, device=dev) * 0.1).to(bf)
    h = torch.randn(M, K, device=dev).to(bf)
    dW, db = torch.zeros(V, K, device=dev), torch.zeros(V, device=dev)
    ops.gemm_tn(dl, h, dW, M, V, K, ld, K, K, colsum=db)
    ref = dl[:, :V].float().t() @ h.float()
    err = ((dW - ref).abs().max() / ref.abs().max()).item()
    t = timeit(lambda: ops.gemm_tn(dl, h, dW, M, V, K, ld, K, K, colsum=db))
    print(f'vocab dW {M} x {V} x {K}: {t:.1f} us  {2.0 * M * V * K / t / 1e6:.0f} TF/s   max-norm error of the first launch {err:.2e}')
