"""The six GEMM launches of a stage-3 / stage-4 MLP (GEMM form), each timed alone with the step's shapes:
fc1 (+bias, GELU, H and G stored) | fc2 (+bias, fp32 residual) | gelu'-dgrad (act 2) | fc1 dgrad | dW2 | dW1."""
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
for M, C, hid in ((98304, 320, 1280), (49152, 512, 2048)):
    x = torch.randn(M, C, device=dev).to(bf)
    w1 = (torch.randn(hid, C, device=dev) * C ** -0.5).to(bf); w2 = (torch.randn(C, hid, device=dev) * hid ** -0.5).to(bf)
    b1 = torch.randn(hid, device=dev) * 0.1; b2 = torch.randn(C, device=dev) * 0.1
    H = torch.empty(M, hid, device=dev, dtype=bf); G = torch.empty_like(H)
    R = torch.randn(M, C, device=dev); out = torch.empty(M, C, device=dev)
    dy = torch.randn(M, C, device=dev).to(bf); dH = torch.empty_like(H); dxn = torch.empty(M, C, device=dev, dtype=bf)
    w2t = w2.t().contiguous(); w1t = w1.t().contiguous()
    dw1 = torch.zeros(hid, C, device=dev); dw2 = torch.zeros(C, hid, device=dev); db1 = torch.zeros(hid, device=dev); db2 = torch.zeros(C, device=dev)
    fl = 2.0 * M * C * hid
    t = [timeit(lambda: ops.gemm_nt(x, w1, G, M, hid, C, C, C, hid, bias=b1, act=1, H=H)),
         timeit(lambda: ops.gemm_nt(G, w2, out, M, C, hid, hid, hid, C, bias=b2, R=R)),
         timeit(lambda: ops.gemm_nt(dy, w2t, dH, M, hid, C, C, C, hid, act=2, H=H)),
         timeit(lambda: ops.gemm_nt(dH, w1t, dxn, M, C, hid, hid, hid, C)),
         timeit(lambda: ops.gemm_tn(dy, G, dw2, M, C, hid, C, hid, hid, colsum=db2)),
         timeit(lambda: ops.gemm_tn(dH, x, dw1, M, hid, C, hid, C, C, colsum=db1))]
    print('M=%d C=%d: ' % (M, C) + ' | '.join('%s %.1f us %.0f TF' % (n, v, fl / v / 1e6) for n, v in zip(('fc1+gelu', 'fc2+R', "gelu'dgrad", 'fc1dgrad', 'dW2', 'dW1'), t)), flush=True)
