"""LayerNorm forward / backward on the four stage shapes (rows x C of the residual stream at batch 256): achieved HBM rate."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16
COPIES = int(os.environ.get('UB_COPIES', '1'))      # interleaved dgamma / dbeta accumulators (FlatStore.LN_COPIES in the model)
def timeit(fn, reps=20):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for rows, C in [(1081344, 64), (294912, 128), (98304, 320), (49152, 512), (1048576, 64), (262144, 128), (65536, 320), (32768, 320), (32768, 768), (16384, 512), (16384, 64)]:
    x = torch.randn(rows, C, device=dev); y = torch.empty(rows, C, device=dev, dtype=bf)
    g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev); mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
    t = timeit(lambda: ops.layernorm_fwd(x, y, g, b, rows, C, C, C, 1e-6, mean=mean, rstd=rstd))
    by = rows * C * 6 + rows * 8
    dy = torch.randn(rows, C, device=dev).to(bf); dx = torch.zeros(rows, C, device=dev); dgc = torch.zeros(COPIES, 1024, device=dev); dbc = torch.zeros(COPIES, 1024, device=dev); dg, db = dgc[0, :C], dbc[0, :C]
    kw = dict(copies=COPIES, copy_stride=1024) if COPIES > 1 else {}
    t2 = timeit(lambda: ops.layernorm_bwd(dy, x, dx, g, mean, rstd, rows, C, C, C, C, dgamma=dg, dbeta=db, accumulate=True, **kw))
    by2 = rows * C * (2 + 4 + 8) + rows * 8
    dxb = torch.empty(rows, C, device=dev, dtype=bf); xb = x.to(bf)
    t3 = timeit(lambda: ops.layernorm_bwd(dy, xb, dxb, g, mean, rstd, rows, C, C, C, C, dgamma=dg, dbeta=db, **kw))
    by3 = rows * C * 6 + rows * 8
    dxa = torch.zeros(rows, C, device=dev, dtype=bf)           # the model's combination: bf16 dy, fp32 x, bf16 gradient stream accumulated in place
    t4 = timeit(lambda: ops.layernorm_bwd(dy, x, dxa, g, mean, rstd, rows, C, C, C, C, dgamma=dg, dbeta=db, accumulate=True, **kw))
    by4 = rows * C * (2 + 4 + 4) + rows * 8
    print('ln rows=%d C=%d: fwd %.1f us %.2f TB/s | bwd(f32 x, dx+=) %.1f us %.2f TB/s | bwd(bf16) %.1f us %.2f TB/s | bwd(bf16 dy, f32 x, bf16 dx+=) %.1f us %.2f TB/s' % (
        rows, C, t * 1e3, by / t / 1e9, t2 * 1e3, by2 / t2 / 1e9, t3 * 1e3, by3 / t3 / 1e9, t4 * 1e3, by4 / t4 / 1e9))
