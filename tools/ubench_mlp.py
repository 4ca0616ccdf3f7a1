import sys, torch
sys.path.insert(0, '.')
from mvlt_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
bf = torch.bfloat16
for C, hid, N in ((64, 512, 4224), (128, 1024, 1152)):
    M = 256 * N
    x = torch.randn(M, C, device=dev).to(bf); dy = torch.randn(M, C, device=dev).to(bf)
    w1 = (torch.randn(hid, C, device=dev) * C ** -0.5).to(bf); w2 = (torch.randn(C, hid, device=dev) * hid ** -0.5).to(bf)
    w1t, w2t = w1.t().contiguous(), w2.t().contiguous()
    b1 = torch.randn(hid, device=dev) * 0.1; b2 = torch.randn(C, device=dev) * 0.1
    res = torch.randn(M, C, device=dev); out = torch.empty_like(res); dx = torch.empty(M, C, device=dev, dtype=bf)
    dw1, db1, dw2, db2 = torch.zeros(hid, C, device=dev), torch.zeros(hid, device=dev), torch.zeros(C, hid, device=dev), torch.zeros(C, device=dev)
    t = timeit(lambda: ops.mlp_fwd(x, w1, b1, w2, b2, res, out, M, C, hid))
    fl = 4.0 * M * C * hid
    print(f'C={C} mlp_fwd   {t:.3f} ms  {fl/t/1e9:.0f} TF/s  alg {(M*C*(2+4+4))/t/1e6:.0f} GB/s')
    t = timeit(lambda: ops.mlp_bwd_dx(x, dy, w1, w1t, w2t, b1, dx, M, C, hid))
    print(f'C={C} mlp_bwd_dx {t:.3f} ms  {1.5*fl/t/1e9:.0f} TF/s')
    t = timeit(lambda: ops.mlp_bwd_dw(x, dy, w1, w2t, b1, dw1, db1, dw2, db2, M, C, hid))
    print(f'C={C} mlp_bwd_dw {t:.3f} ms  {2*fl/t/1e9:.0f} TF/s')
