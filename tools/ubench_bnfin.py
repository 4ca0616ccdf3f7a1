"""BatchNorm finalize + normalise in one launch (mvlt_bn_finalize_norm) at the MIM decoder's shapes; MVLT_BN_FIN_CAP = workgroup cap of the launch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=30):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for M, C in ((262144, 64), (262144, 128), (262144, 192), (65536, 128), (65536, 192), (16384, 64)):
    z = torch.randn(M, C, device=dev).to(torch.float16)
    s1, s2 = torch.randn(16, C, device=dev), torch.rand(16, C, device=dev) * M
    mean, rstd, rm, rv, gamma, beta = (torch.zeros(C, device=dev) for _ in range(6))
    y16 = torch.empty(M, C, device=dev, dtype=torch.bfloat16); y32 = torch.empty(M, C, device=dev)
    t16 = timeit(lambda: ops.bn_finalize_norm(z, C, s1, s2, 16, 1e-5, 0.1, mean, rstd, rm, rv, gamma, beta, M, C, y16=y16, ld16=C))
    t32 = timeit(lambda: ops.bn_finalize_norm(z, C, s1, s2, 16, 1e-5, 0.1, mean, rstd, rm, rv, gamma, beta, M, C, y32=y32, ld32=C))
    print(f"M={M} C={C}: -> bf16 {t16:6.1f} us ({M*C*4/t16/1e6:.2f} TB/s)   -> f32 {t32:6.1f} us ({M*C*6/t32/1e6:.2f} TB/s)")
