"""Does the 256 MB Infinity Cache serve a producer -> consumer chain of the stage-3 MLP when the rows are processed in chunks?
fc1 (+bias, GELU, H and G stored) -> fc2 (+bias, residual fp32) over M = 98304 rows at once, and the same two launches over 2 / 4 / 8 row chunks
back to back (each chunk's G is read right after it was written: 252 / n MB).  Same kernels, same bytes; only the distance between write and read changes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16
def timeit(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for M, C, hid in ((98304, 320, 1280), (49152, 512, 2048)):
    x = torch.randn(M, C, device=dev).to(bf)
    w1 = (torch.randn(hid, C, device=dev) * C ** -0.5).to(bf); w2 = (torch.randn(C, hid, device=dev) * hid ** -0.5).to(bf)
    b1 = torch.randn(hid, device=dev) * 0.1; b2 = torch.randn(C, device=dev) * 0.1
    H = torch.empty(M, hid, device=dev, dtype=bf); G = torch.empty_like(H)
    R = torch.randn(M, C, device=dev); out = torch.empty(M, C, device=dev)
    dy = torch.randn(M, C, device=dev).to(bf); dH = torch.empty_like(H); dxn = torch.empty(M, C, device=dev, dtype=bf)
    w2t = w2.t().contiguous(); w1t = w1.t().contiguous()
    def fwd(n):
        step = M // n
        for i in range(n):
            s = slice(i * step, (i + 1) * step)
            ops.gemm_nt(x[s], w1, G[s], step, hid, C, C, C, hid, bias=b1, act=1, H=H[s])
            ops.gemm_nt(G[s], w2, out[s], step, C, hid, hid, hid, C, bias=b2, R=R[s])
    def bwd(n):                      # act2 (dgrad fc2 * gelu'(H)) -> dgrad fc1
        step = M // n
        for i in range(n):
            s = slice(i * step, (i + 1) * step)
            ops.gemm_nt(dy[s], w2t, dH[s], step, hid, C, C, C, hid, act=2, H=H[s])
            ops.gemm_nt(dH[s], w1t, dxn[s], step, C, hid, hid, hid, C)
    for n in (1, 2, 4, 8):
        print('M=%d C=%d chunks=%d: fc1->fc2 %.1f us | act2->dgrad %.1f us' % (M, C, n, timeit(lambda: fwd(n)) * 1e3, timeit(lambda: bwd(n)) * 1e3), flush=True)
