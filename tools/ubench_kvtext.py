import os, sys, torch
sys.path.insert(0, '/root/repo')
from mvlt_amd import ops
from mvlt_amd._lib import rowmap
dev = torch.device('cuda:0'); bf = torch.bfloat16
def timeit(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
B, T, N, HW, C, Mk, HWr = 256, 128, 4224, 4096, 64, 192, 64
x = torch.randn(B, N, C, device=dev).to(bf); w = (torch.randn(2 * C, C, device=dev) * 0.1).to(bf); b = torch.randn(2 * C, device=dev)
kv = torch.empty(B, Mk, 2 * C, device=dev, dtype=bf)
xt = torch.randn(B * T, C, device=dev).to(bf); out = torch.empty(B * T, 2 * C, device=dev, dtype=bf)
print('plain       %.1f us' % timeit(lambda: ops.gemm_nt(xt, w, out, B * T, 2 * C, C, C, C, 2 * C, bias=b)))
print('A map       %.1f us' % timeit(lambda: ops.gemm_nt(x, w, out, B * T, 2 * C, C, C, C, 2 * C, bias=b, a_map=rowmap(T, N, HW))))
print('C map       %.1f us' % timeit(lambda: ops.gemm_nt(xt, w, kv, B * T, 2 * C, C, C, C, 2 * C, bias=b, c_map=rowmap(T, Mk, HWr))))
print('A and C map %.1f us' % timeit(lambda: ops.gemm_nt(x, w, kv, B * T, 2 * C, C, C, C, 2 * C, bias=b, a_map=rowmap(T, N, HW), c_map=rowmap(T, Mk, HWr))))
print('no bias A+C %.1f us' % timeit(lambda: ops.gemm_nt(x, w, kv, B * T, 2 * C, C, C, C, 2 * C, a_map=rowmap(T, N, HW), c_map=rowmap(T, Mk, HWr))))
