"""Do a weight-gradient (TN) launch and an independent input-gradient (NT) launch overlap when issued on two HIP streams?  (stage 3-4 MLP shapes)
    gpurun -- python tools/ubench_streams.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
dev = torch.device('cuda:0'); bf = torch.bfloat16
side = torch.cuda.Stream()
for M, hid, C in ((98304, 1280, 320), (49152, 2048, 512)):
    dy = torch.randn(M, C, device=dev).to(bf); g = torch.randn(M, hid, device=dev).to(bf); H = torch.randn(M, hid, device=dev).to(bf)
    w2t = (torch.randn(hid, C, device=dev) * 0.05).to(bf); w1t = (torch.randn(C, hid, device=dev) * 0.05).to(bf)
    dW2, dW1 = torch.zeros(C, hid, device=dev), torch.zeros(hid, C, device=dev)
    dh = torch.empty(M, hid, device=dev, dtype=bf); xn = torch.randn(M, C, device=dev).to(bf); dxn = torch.empty(M, C, device=dev, dtype=bf)
    cs2, cs1 = torch.zeros(C, device=dev), torch.zeros(hid, device=dev)

    def seq():
        ops.gemm_tn(dy, g, dW2, M, C, hid, C, hid, hid, colsum=cs2)
        ops.gemm_nt(dy, w2t, dh, M, hid, C, C, C, hid, act=2, H=H)
        ops.gemm_tn(dh, xn, dW1, M, hid, C, hid, C, C, colsum=cs1)
        ops.gemm_nt(dh, w1t, dxn, M, C, hid, hid, hid, C)

    def par():
        main = torch.cuda.current_stream()
        e0 = torch.cuda.Event(); e0.record(main); side.wait_event(e0)
        with torch.cuda.stream(side):
            ops.gemm_tn(dy, g, dW2, M, C, hid, C, hid, hid, colsum=cs2)
        ops.gemm_nt(dy, w2t, dh, M, hid, C, C, C, hid, act=2, H=H)
        e1 = torch.cuda.Event(); e1.record(main); side.wait_event(e1)
        with torch.cuda.stream(side):
            ops.gemm_tn(dh, xn, dW1, M, hid, C, hid, C, C, colsum=cs1)
        ops.gemm_nt(dh, w1t, dxn, M, C, hid, hid, hid, C)
        main.wait_stream(side)

    def timeit(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3
    print(f"M={M} hid={hid} C={C}: one stream {timeit(seq):.1f} us, weight gradients on a side stream {timeit(par):.1f} us")
