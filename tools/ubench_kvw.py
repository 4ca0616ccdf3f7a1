import os, sys, torch
sys.path.insert(0, '/root/repo')
from mvlt_amd import ops
from mvlt_amd._lib import rowmap
dev = torch.device('cuda:0'); bf = torch.bfloat16
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
B, T = 256, 128
for C, HWr, N in [(128, 64, 1152), (320, 64, 384), (64, 64, 4224)]:
    Mk = HWr + T
    dkv = torch.randn(B, Mk, 2 * C, device=dev).to(bf)
    xn1 = torch.randn(B, N, C, device=dev).to(bf)
    kvin = torch.randn(B * HWr, C, device=dev).to(bf)
    g = torch.zeros(2 * C, C, device=dev); gb = torch.zeros(2 * C, device=dev)
    HW = N - T
    for sp in (0, 8, 16, 32, 64, 128):
        t1 = timeit(lambda: ops.gemm_tn(dkv, xn1, g, B * T, 2 * C, C, 2 * C, C, C, a_map=rowmap(T, Mk, HWr), b_map=rowmap(T, N, HW), colsum=gb, splits=sp))
        t2 = timeit(lambda: ops.gemm_tn(dkv, kvin, g, B * HWr, 2 * C, C, 2 * C, C, C, a_map=rowmap(HWr, Mk, 0), colsum=gb, splits=sp))
        print(f'C={C} splits={sp}: text rows {t1*1e3:.1f} us  image rows {t2*1e3:.1f} us')
    # merged: one [B*Mk] row GEMM against a packed kv_in buffer
    kvall = torch.randn(B * Mk, C, device=dev).to(bf)
    for sp in (0, 16, 32, 64):
        t3 = timeit(lambda: ops.gemm_tn(dkv, kvall, g, B * Mk, 2 * C, C, 2 * C, C, C, colsum=gb, splits=sp))
        print(f'C={C} splits={sp}: merged {t3*1e3:.1f} us')
