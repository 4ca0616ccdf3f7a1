"""The weight-gradient GEMMs that reduce through bf16 partial tiles + an ordered fold (default when a scratch is passed) against the fp32 atomics (MVLT_TN_P8=0),
on the stage-4 / stage-3 MLP weight-gradient shapes: accuracy against an fp32 reference, bit-identical second launch, time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
from mvlt_amd._lib import last_kernel
dev = torch.device('cuda:0'); bf = torch.bfloat16
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for M, N1, N2 in ((49152, 512, 2048), (49152, 2048, 512), (49152, 512, 512), (49152, 1024, 512), (98304, 1280, 320), (98304, 320, 1280)):
    A = (torch.randn(M, N1, device=dev) * 0.5).to(bf); B = (torch.randn(M, N2, device=dev) * 0.5).to(bf)
    ref = A.float().t() @ B.float()
    scr = torch.empty(256 * 65536, device=dev, dtype=bf)      # mvlt_gemm_tn_args.partials (MVLT_TN_P8=0: ignored, fp32 atomics)
    outs = []
    for rep in range(2):
        Cw, cs = torch.zeros(N1, N2, device=dev), torch.zeros(N1, device=dev)
        ops.gemm_tn(A, B, Cw, M, N1, N2, N1, N2, N2, colsum=cs, partials=scr)
        torch.cuda.synchronize()
        outs.append((Cw.clone(), cs.clone()))
    kern = last_kernel()
    err = ((outs[0][0] - ref).abs().max() / ref.abs().max()).item()
    erb = ((outs[0][1] - A.float().sum(0)).abs().max() / A.float().sum(0).abs().max()).item()
    same = bool(torch.equal(outs[0][0], outs[1][0]))
    Cw, cs = torch.zeros(N1, N2, device=dev), torch.zeros(N1, device=dev)
    t = timeit(lambda: ops.gemm_tn(A, B, Cw, M, N1, N2, N1, N2, N2, colsum=cs, partials=scr))
    print(f"{M} x {N1} x {N2}: {t:.1f} us  {2.0 * M * N1 * N2 / t / 1e6:.0f} TF/s  max-norm error {err:.2e} (bias gradient {erb:.1e})  second launch bit-identical: {same}   [{kern}]")
