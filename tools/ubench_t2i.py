"""MIM loss tail: (x8 upsample -> SmoothL1 fwd; SmoothL1 bwd -> upsample bwd) against the fused pair that never writes the prediction."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
B, H, s = 256, 32, 8
sc = torch.randn(B * H * H, 3, device=dev); target = torch.rand(B, 3, H * s, H * s, device=dev)
out = torch.empty(B, 3, H * s, H * s, device=dev); acc = torch.zeros(1, device=dev); g = torch.ones(1, device=dev)
grad = torch.empty_like(out); dsc = torch.zeros(B * H * H, 8, device=dev, dtype=torch.bfloat16)
print('upsample_fwd %.1f us' % timeit(lambda: ops.upsample_fwd(sc, 3, B, H, H, 3, s, out, 0, nchw=True)))
print('smooth_l1_fwd %.1f us' % timeit(lambda: ops.smooth_l1_fwd(out, target, acc)))
print('smooth_l1_bwd %.1f us' % timeit(lambda: ops.smooth_l1_bwd(out, target, g, grad)))
print('upsample_bwd %.1f us' % timeit(lambda: ops.upsample_bwd(grad, 0, True, B, H, H, 3, s, dsc, 8)))
print('fused fwd %.1f us' % timeit(lambda: ops.upsample_l1_fwd(sc, 3, B, H, H, 3, s, target, acc)))
print('fused bwd %.1f us' % timeit(lambda: ops.upsample_l1_bwd(sc, 3, B, H, H, 3, s, target, g, dsc, 8)))
