"""A/B of the fused-MLP kernels inside one process: MVLT_MLP_LEGACY is read once per process by the library, so this script is run
twice by its caller (tools/ab_mlp.sh) -- or use h_out to force the legacy forward."""
import os, sys, torch
sys.path.insert(0, '.')
from mvlt_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
bf = torch.bfloat16
tag = "legacy" if os.environ.get("MVLT_MLP_LEGACY") else "pipe"
for C, hid, N in ((64, 512, 4224), (128, 1024, 1152)):
    M = 256 * N
    x = torch.randn(M, C, device=dev).to(bf); dy = torch.randn(M, C, device=dev).to(bf)
    w1 = (torch.randn(hid, C, device=dev) * C ** -0.5).to(bf); w2 = (torch.randn(C, hid, device=dev) * hid ** -0.5).to(bf)
    w1t, w2t = w1.t().contiguous(), w2.t().contiguous()
    b1 = torch.randn(hid, device=dev) * 0.1; b2 = torch.randn(C, device=dev) * 0.1
    res = torch.randn(M, C, device=dev); out = torch.empty_like(res); dx = torch.empty(M, C, device=dev, dtype=bf)
    dw1, db1, dw2, db2 = torch.zeros(hid, C, device=dev), torch.zeros(hid, device=dev), torch.zeros(C, hid, device=dev), torch.zeros(C, device=dev)
    fl = 4.0 * M * C * hid
    for rep in range(2):
        t = timeit(lambda: ops.mlp_fwd(x, w1, b1, w2, b2, res, out, M, C, hid))
        print(f'{tag} C={C} mlp_fwd    {t*1e3:7.1f} us  {fl/t/1e9:.0f} TF/s', flush=True)
        t = timeit(lambda: ops.mlp_bwd_dx(x, dy, w1, w1t, w2t, b1, dx, M, C, hid))
        print(f'{tag} C={C} mlp_bwd_dx {t*1e3:7.1f} us  {1.5*fl/t/1e9:.0f} TF/s', flush=True)
        t = timeit(lambda: ops.mlp_bwd_dw(x, dy, w1, w2t, b1, dw1, db1, dw2, db2, M, C, hid))
        print(f'{tag} C={C} mlp_bwd_dw {t*1e3:7.1f} us  {2*fl/t/1e9:.0f} TF/s', flush=True)
