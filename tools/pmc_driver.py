#!/usr/bin/env python3
"""Launch the hot kernels of the step alone (3x each, bench-sized shapes, random data) so that a rocprofv3 --pmc pass sees only them:
the fused-MLP family (stage 1 and 2), the roofline GEMM (MIM conv3x3 192->192 as gathered GEMM), the stage-1 K=64 projection, the
stage-3 fc1 GEMM with the GELU epilogue, its weight-gradient GEMM, and the stage-1 attention forward / backward.

    rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d gpurun_out/pmc_x -o x -- python3 tools/pmc_driver.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mvlt_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    bf = torch.bfloat16
    B = int(os.environ.get("B", "256"))
    todo = []
    for C, hid, N in ((64, 512, 4224), (128, 1024, 1152)):
        M = B * N
        x, dy = torch.randn(M, C, device=dev).to(bf), torch.randn(M, C, device=dev).to(bf)
        w1, w2 = (torch.randn(hid, C, device=dev) * C ** -0.5).to(bf), (torch.randn(C, hid, device=dev) * hid ** -0.5).to(bf)
        w1t, w2t = w1.t().contiguous(), w2.t().contiguous()
        b1, b2 = torch.randn(hid, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1
        res = torch.randn(M, C, device=dev)
        out, dx = torch.empty_like(res), torch.empty(M, C, device=dev, dtype=bf)
        dw1, db1, dw2, db2 = torch.zeros(hid, C, device=dev), torch.zeros(hid, device=dev), torch.zeros(C, hid, device=dev), torch.zeros(C, device=dev)
        todo.append(lambda x=x, w1=w1, b1=b1, w2=w2, b2=b2, res=res, out=out, M=M, C=C, hid=hid: ops.mlp_fwd(x, w1, b1, w2, b2, res, out, M, C, hid))
        todo.append(lambda x=x, dy=dy, w1=w1, w1t=w1t, w2t=w2t, b1=b1, dx=dx, M=M, C=C, hid=hid: ops.mlp_bwd_dx(x, dy, w1, w1t, w2t, b1, dx, M, C, hid))
        todo.append(lambda x=x, dy=dy, w1=w1, w2t=w2t, b1=b1, a=(dw1, db1, dw2, db2), M=M, C=C, hid=hid: ops.mlp_bwd_dw(x, dy, w1, w2t, b1, *a, M, C, hid))
    for _, fn, _ in bench.roofline_cases(B, dev):
        todo.append(fn)
    # stage-3 fc1 (+bias, GELU, pre-activation store) and its weight gradient
    M, N, K = B * 384, 1280, 320
    a, w = torch.randn(M, K, device=dev).to(bf), (torch.randn(N, K, device=dev) * K ** -0.5).to(bf)
    bias, o, h = torch.randn(N, device=dev), torch.empty(M, N, device=dev, dtype=bf), torch.empty(M, N, device=dev, dtype=bf)
    todo.append(lambda: ops.gemm_nt(a, w, o, M, N, K, K, K, N, bias=bias, act=1, H=h))
    dwg = torch.zeros(N, K, device=dev)
    todo.append(lambda: ops.gemm_tn(o, a, dwg, M, N, K, N, K, K))
    # round 4: the other launches of the 8-wave / 8-phase NT kernels -- stage-3 GELU' dgrad (256 x 256, EPI 4), fc2 + fp32 residual and fc1 dgrad
    # (192 x 320 tiles, K = 1280), stage-4 fc1 dgrad (192 x 256, K = 2048) -- and the fused weight + input gradient of a stage-1 C x C Linear
    dyh = torch.randn(M, K, device=dev).to(bf)
    w2t = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf)
    dh = torch.empty(M, N, device=dev, dtype=bf)
    todo.append(lambda: ops.gemm_nt(dyh, w2t, dh, M, N, K, K, K, N, act=2, H=h))
    w2 = (torch.randn(K, N, device=dev) * N ** -0.5).to(bf)
    b2, r32, o32 = torch.randn(K, device=dev), torch.randn(M, K, device=dev), torch.empty(M, K, device=dev)
    todo.append(lambda: ops.gemm_nt(o, w2, o32, M, K, N, N, N, K, bias=b2, R=r32))
    dxn = torch.empty(M, K, device=dev, dtype=bf)
    todo.append(lambda: ops.gemm_nt(dh, w2, dxn, M, K, N, N, N, K))
    M4, C4, H4 = B * 192, 512, 2048
    dh4, w14 = torch.randn(M4, H4, device=dev).to(bf), (torch.randn(C4, H4, device=dev) * H4 ** -0.5).to(bf)
    dx4 = torch.empty(M4, C4, device=dev, dtype=bf)
    todo.append(lambda: ops.gemm_nt(dh4, w14, dx4, M4, C4, H4, H4, H4, C4))
    # round 5: the stage-4 fc2 weight gradient on the 8-phase TN loop with bf16 partial tiles + fold (gemm_tn_p8_kernel, tn_fold_kernel), and the same shape on the atomic kernel
    dy4, g4 = torch.randn(M4, C4, device=dev).to(bf), torch.randn(M4, H4, device=dev).to(bf)
    dw4, db4, scr = torch.zeros(C4, H4, device=dev), torch.zeros(C4, device=dev), torch.empty(256 * 65536, device=dev, dtype=bf)
    todo.append(lambda: ops.gemm_tn(dy4, g4, dw4, M4, C4, H4, C4, H4, H4, colsum=db4, partials=scr))
    todo.append(lambda: ops.gemm_tn(dy4, g4, dw4, M4, C4, H4, C4, H4, H4, colsum=db4))
    M1 = B * 4224
    dq, xn = torch.randn(M1, 64, device=dev).to(bf), torch.randn(M1, 64, device=dev).to(bf)
    wqt = (torch.randn(64, 64, device=dev) * 0.125).to(bf)
    dwq, dbq, dxq = torch.zeros(64, 64, device=dev), torch.zeros(64, device=dev), torch.empty(M1, 64, device=dev, dtype=bf)
    todo.append(lambda: ops.gemm_tn(dq, xn, dwq, M1, 64, 64, 64, 64, 64, colsum=dbq, dgrad=(wqt, dxq)))
    # MIM decoder: weight gradient of the 192 -> 192 conv3x3 at 32 x 32 (conv3_wgrad_kernel: LDS-resident halo)
    from mvlt_amd._lib import conv3map
    Mc, Cc2 = B * 1024, 192
    xz, dz = torch.randn(B, 1024, Cc2, device=dev).to(bf), torch.randn(Mc, Cc2, device=dev).to(bf)
    dwc = torch.zeros(Cc2, 9 * Cc2, device=dev)
    cmap = conv3map(32, 32, 1024, Cc2)
    todo.append(lambda: ops.gemm_tn(dz, xz, dwc, Mc, Cc2, 9 * Cc2, Cc2, Cc2, 9 * Cc2, b_map=cmap))
    # stage-1 attention: B x 1 head, 4224 queries, 192 keys
    Nq, Mk, Cc = 4224, 192, 64
    q, kv = torch.randn(B, Nq, Cc, device=dev).to(bf), torch.randn(B, Mk, 2 * Cc, device=dev).to(bf)
    ao, lse = torch.empty_like(q), torch.empty(B, 1, Nq, device=dev)
    do, dq, dkv = torch.randn(B, Nq, Cc, device=dev).to(bf), torch.empty_like(q), torch.zeros(B, Mk, 2 * Cc, device=dev)
    todo.append(lambda: ops.sr_attention_fwd(q, kv, ao, lse, B, 1, Nq, Mk, Cc, 2 * Cc, Cc, 0, Cc, 0.125))
    todo.append(lambda: ops.sr_attention_bwd(q, kv, ao, do, lse, dq, dkv, B, 1, Nq, Mk, Cc, 2 * Cc, Cc, 2 * Cc, 0, Cc, 0.125))
    for fn in todo:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
