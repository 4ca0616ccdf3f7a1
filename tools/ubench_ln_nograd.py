import os, sys, torch
sys.path.insert(0, '/root/repo')
from mvlt_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16
def timeit(fn, reps=20):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for rows, C in [(98304, 320), (65536, 320), (32768, 320), (32768, 768), (16384, 512), (49152, 512)]:
    x = torch.randn(rows, C, device=dev); g = torch.ones(C, device=dev); mean = torch.zeros(rows, device=dev); rstd = torch.ones(rows, device=dev)
    dy = torch.randn(rows, C, device=dev).to(bf); dx = torch.zeros(rows, C, device=dev)
    NC = int(os.environ.get("UB_COPIES", "8")); dgc = torch.zeros(NC, 1024, device=dev); dbc = torch.zeros(NC, 1024, device=dev); dg, db = dgc[0, :C], dbc[0, :C]
    xb = x.to(bf); dxb = torch.empty(rows, C, device=dev, dtype=bf)
    t1 = timeit(lambda: ops.layernorm_bwd(dy, xb, dxb, g, mean, rstd, rows, C, C, C, C, dgamma=dg, dbeta=db, copies=NC, copy_stride=1024))
    t2 = timeit(lambda: ops.layernorm_bwd(dy, xb, dxb, g, mean, rstd, rows, C, C, C, C))
    t3 = timeit(lambda: ops.layernorm_bwd(dy, x, dx, g, mean, rstd, rows, C, C, C, C, dgamma=dg, dbeta=db, accumulate=True, copies=NC, copy_stride=1024))
    t4 = timeit(lambda: ops.layernorm_bwd(dy, x, dx, g, mean, rstd, rows, C, C, C, C, accumulate=True))
    print(f"rows={rows} C={C}: bf16 with dgamma {t1:.1f} without {t2:.1f} | f32 acc with {t3:.1f} without {t4:.1f}")
