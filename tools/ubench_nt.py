"""NT GEMM microbench on the stage-3/4 MLP shapes (also the target of rocprofv3 --pmc passes)."""
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
for M, N, K in [(98304, 1280, 320), (49152, 2048, 512), (98304, 320, 1280), (49152, 512, 2048), (262144, 256, 1728)]:
    A = torch.randn(M, K, device=dev).to(bf); W = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf)
    b = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev, dtype=bf); H = torch.empty_like(out)
    for name, kw in [('plain', {}), ('bias+gelu', dict(bias=b, act=1)), ('bias+gelu+H', dict(bias=b, act=1, H=H)), ("gelu'(H)", dict(act=2, H=H))]:
        t = timeit(lambda: ops.gemm_nt(A, W, out, M, N, K, K, K, N, **kw))
        nout = 2 if 'H' in kw and kw.get('act') == 1 else 1
        by = (M * K + N * K + nout * M * N + (M * N if kw.get('act') == 2 else 0)) * 2
        print('nt M=%d N=%d K=%d %-12s %.1f us  %.0f TF/s  %.0f GB/s' % (M, N, K, name, t * 1e3, 2.0 * M * N * K / t / 1e9, by / t / 1e6))
