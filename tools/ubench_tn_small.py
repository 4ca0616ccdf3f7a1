"""The q / proj / kv weight-gradient shapes of stages 3-4 (outputs of 320 x 320 .. 1024 x 512) alone: how much of the launch is the split reduction?
A/B against an ablation build without the output atomics: MVLT_HIP_LIB=ab/libmvlt_tnnoat.so (tools/build_alt.sh tnnoat gemm.hip -DMVLT_TN_NOATOMIC=1)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for M, N1, N2 in ((98304, 320, 320), (49152, 512, 512), (49152, 1024, 512), (98304, 1280, 320), (98304, 320, 1280), (294912, 128, 128), (262144, 192, 192)):
    A = (torch.randn(M, N1, device=dev) * 0.5).to(bf); B = (torch.randn(M, N2, device=dev) * 0.5).to(bf)
    Cw, cs = torch.zeros(N1, N2, device=dev), torch.zeros(N1, device=dev)
    scr = torch.empty(256 * 65536, device=dev, dtype=bf)
    t = timeit(lambda: ops.gemm_tn(A, B, Cw, M, N1, N2, N1, N2, N2, colsum=cs, partials=scr))
    print(f"{M} x {N1} x {N2}: {t:.1f} us  {2.0 * M * N1 * N2 / t / 1e6:.0f} TF/s")
