"""conv3x3 weight-gradient microbench on the eleven MIM-decoder convolutions of pvlt_tiny at batch 256 (256 px):
    MVLT_NO_CONV_WGRAD=1 python tools/ubench_convw.py     # generic gathered TN GEMM
    python tools/ubench_convw.py                          # conv3_wgrad_kernel (LDS halo)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
from mvlt_amd._lib import conv3map
dev = torch.device('cuda:0'); bf = torch.bfloat16
def timeit(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
B, tot = 256, 0.0
# (side, cin, cout, tokens_in, count per step)
for side, cin, cout, tok, cnt in ((32, 192, 192, 1024, 2), (32, 64, 64, 1024, 2), (32, 128, 128, 1024, 1), (32, 128, 64, 1152, 1),
                                  (16, 64, 64, 256, 2), (16, 128, 128, 256, 1), (16, 320, 64, 384, 1), (8, 512, 64, 192, 1)):
    M = B * side * side
    x = torch.randn(B, tok, cin, device=dev).to(bf); dz = torch.randn(M, cout, device=dev).to(bf)
    dW = torch.zeros(cout, 9 * cin, device=dev)
    scr = torch.empty(512 * 65536, device=dev, dtype=bf)      # mvlt_gemm_tn_args.partials: bf16 partial blocks + ordered fold (MVLT_TN_P8=0: fp32 atomics)
    t = timeit(lambda: ops.gemm_tn(dz, x, dW, M, cout, 9 * cin, cout, cin, 9 * cin, b_map=conv3map(side, side, tok, cin), partials=scr))
    tot += cnt * t
    print(f'{side}x{side} {cin:>3}->{cout:<3} x{cnt}: {t*1e3:7.1f} us  {2.0*M*cout*9*cin/t/1e9:6.0f} TF/s')
print(f'per step: {tot:.3f} ms')
