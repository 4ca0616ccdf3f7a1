"""NT GEMM microbench on the short-K shapes (K <= 512): python tools/ubench_nt2.py [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
def timeit(fn):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for M, N, K in [(262144, 192, 1728), (98304, 320, 320), (49152, 512, 512), (49152, 1024, 512), (294912, 128, 128), (294912, 256, 128), (98304, 640, 320), (262144, 192, 576), (262144, 128, 1152)]:
    A = torch.randn(M, K, device=dev).to(bf); W = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf)
    b = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev, dtype=bf); R = torch.randn(M, N, device=dev); o32 = torch.empty_like(R)
    rs = torch.ones(M // 64, device=dev)
    for name, o, kw in [('plain', out, {}), ('bias', out, dict(bias=b)), ('bias*rs+R f32', o32, dict(bias=b, R=R, row_scale=rs, rows_per_scale=64))]:
        t = timeit(lambda: ops.gemm_nt(A, W, o, M, N, K, K, K, N, **kw))
        print('nt M=%d N=%d K=%d %-14s %.1f us  %.0f TF/s' % (M, N, K, name, t * 1e3, 2.0 * M * N * K / t / 1e9))
