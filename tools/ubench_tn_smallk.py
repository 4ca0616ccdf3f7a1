"""Small weight-gradient GEMMs of the step (few k-tiles per workgroup, narrow outputs): time against the HBM time of their operands, by number of m-splits.
    python tools/ubench_tn_smallk.py            (MVLT_TN_MINT = minimum k-tiles per split)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
from mvlt_amd._lib import last_kernel
dev = torch.device('cuda:0'); bf = torch.bfloat16
scr = torch.empty(64 * 8 * 65536, dtype=bf, device=dev)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
SHAPES = [(32768, 64, 768), (32768, 256, 128), (16384, 256, 128), (32768, 320, 128), (1048576, 64, 48), (32768, 512, 320), (16384, 640, 320), (32768, 640, 320), (32768, 128, 64)]
for M, N1, N2 in SHAPES:
    A, B = torch.randn(M, N1, device=dev).to(bf), torch.randn(M, N2, device=dev).to(bf)
    out, cs = torch.zeros(N1, N2, device=dev), torch.zeros(N1, device=dev)
    row = []
    for splits in (0, 16, 32, 64, 128, 256, 512):
        t = timeit(lambda: ops.gemm_tn(A, B, out, M, N1, N2, N1, N2, N2, colsum=cs, splits=splits))
        row.append(f"{splits}:{t:.1f}")
    for splits in (0, 32, 64, 128, 256):               # the same with the partial-tile scratch (bf16 partial tiles + an immediate fold, timed with it)
        t = timeit(lambda: ops.gemm_tn(A, B, out, M, N1, N2, N1, N2, N2, colsum=cs, splits=splits, partials=scr))
        row.append(f"p{splits}:{t:.1f}")
    floor = (A.numel() + B.numel()) * 2 / 6.3e6
    print(f"tn {M} x {N1} x {N2}: " + "  ".join(row) + f"   us  (operands at 6.3 TB/s: {floor:.1f} us)  [{last_kernel()}]")
