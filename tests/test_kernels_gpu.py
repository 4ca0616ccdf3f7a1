"""GPU: every HIP kernel, called through the C ABI, against a plain PyTorch fp32 reference of the same op.
Tolerances: fp32 instantiations 1e-3 relative (north_star fp32 bar), bf16 2e-2 (bf16 bar)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 1e-3, torch.bfloat16: 2e-2}


def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def maxrel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def rnd(*shape, dtype, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dev()).to(dtype)


@pytest.fixture(scope="module")
def ops():
    from mvlt_amd import ops as _ops
    return _ops


# ------------------------------------------------------------------ gemm_nt
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K", [(300, 64, 64), (1000, 512, 64), (257, 320, 1280), (130, 30522, 768), (64, 2, 768),
                                   (129, 48, 48), (4224 * 2, 64, 512), (5, 122, 768), (1000, 384, 256), (777, 192, 1728)])
def test_gemm_nt_plain(ops, dtype, M, N, K):
    A, B = rnd(M, K, dtype=dtype), rnd(N, K, dtype=dtype, seed=1)
    bias = rnd(N, dtype=torch.float32, seed=2)
    out = torch.empty(M, N, device=dev(), dtype=dtype)
    ops.gemm_nt(A, B, out, M, N, K, K, K, N, bias=bias)
    ref = A.float() @ B.float().t() + bias
    assert maxrel(out.float(), ref) < TOL[dtype], (M, N, K)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("N", [256, 192, 320])        # 192 / 320 run as a 128-wide + a 64-wide launch (ops.gemm_nt)
def test_gemm_nt_epilogues(ops, dtype, N):
    M, K, Bsz = 384, 128, 3
    A, W = rnd(M, K, dtype=dtype), rnd(N, K, dtype=dtype, seed=1, scale=0.2)
    bias = rnd(N, dtype=torch.float32, seed=2)
    ref_pre = A.float() @ W.float().t() + bias
    # gelu with saved pre-activation
    out = torch.empty(M, N, device=dev(), dtype=dtype)
    H = torch.empty_like(out)
    ops.gemm_nt(A, W, out, M, N, K, K, K, N, bias=bias, act=1, H=H)
    assert maxrel(H.float(), ref_pre) < TOL[dtype]
    assert maxrel(out.float(), F.gelu(ref_pre)) < TOL[dtype]
    # gelu' multiply
    Hin = rnd(M, N, dtype=dtype, seed=5)
    out2 = torch.empty_like(out)
    ops.gemm_nt(A, W, out2, M, N, K, K, K, N, act=2, H=Hin)
    h = Hin.float().requires_grad_(True)
    F.gelu(h).sum().backward()
    assert maxrel(out2.float(), (A.float() @ W.float().t()) * h.grad) < TOL[dtype]
    # residual + per-sample row scale, in place (R aliases C)
    Rres = rnd(M, N, dtype=dtype, seed=7)
    scale = torch.tensor([0.0, 1.0 / 0.9, 1.0], device=dev())
    out3 = Rres.clone()
    ops.gemm_nt(A, W, out3, M, N, K, K, K, N, bias=bias, row_scale=scale, rows_per_scale=M // Bsz, R=out3)
    ref3 = ref_pre * scale.repeat_interleave(M // Bsz)[:, None] + Rres.float()
    assert maxrel(out3.float(), ref3) < TOL[dtype]
    # fp32 output from bf16 operands
    out4 = torch.empty(M, N, device=dev(), dtype=torch.float32)
    ops.gemm_nt(A, W, out4, M, N, K, K, K, N, bias=bias)
    assert maxrel(out4, ref_pre) < TOL[dtype]
    if dtype == torch.bfloat16 and N % 8 == 0:
        # fp32 residual beside a bf16 C (mvlt_gemm_nt_args.r_fp32) == the all-fp32 epilogue's result rounded once
        R32 = rnd(M, N, dtype=torch.float32, seed=9)
        o32, o16 = torch.empty(M, N, device=dev(), dtype=torch.float32), torch.empty(M, N, device=dev(), dtype=dtype)
        ops.gemm_nt(A, W, o32, M, N, K, K, K, N, bias=bias, row_scale=scale, rows_per_scale=M // Bsz, R=R32)
        ops.gemm_nt(A, W, o16, M, N, K, K, K, N, bias=bias, row_scale=scale, rows_per_scale=M // Bsz, R=R32)
        assert torch.equal(o16, o32.to(dtype))


@pytest.mark.parametrize("M,N,K,ldc", [(1490, 30522, 768, 30528), (257, 98309, 128, 98312), (1024, 24580, 192, 24584), (1490, 3002, 768, 3008)])
@pytest.mark.parametrize("odt", [torch.float32, torch.bfloat16])
def test_gemm_nt_ragged_rows_and_columns(ops, M, N, K, ldc, odt):
    """the MLM logits shape: neither M nor N whole tiles, N not even a multiple of 8 (row stride padded to one).  From 384 tiles on this is the ragged
    form of the 8-phase kernel (loader clamped at the last row, checked epilogue); the last chunk of a row is stored column by column, so the
    padding columns of the buffer keep what they held.  The last case stays on the 128-wide kernel (same epilogue)."""
    dt = torch.bfloat16
    A, W = rnd(M, K, dtype=dt), rnd(N, K, dtype=dt, seed=1, scale=0.2)
    bias = rnd(N, dtype=torch.float32, seed=2)
    ref = A.float() @ W.float().t() + bias
    buf = torch.full((M + 1, ldc), 7.0, device=dev(), dtype=odt)          # one guard row behind the last
    ops.gemm_nt(A, W, buf, M, N, K, K, K, ldc, bias=bias)
    assert maxrel(buf[:M, :N].float(), ref) < TOL[dt]
    assert (buf[:M, N:] == 7.0).all() and (buf[M] == 7.0).all(), "wrote outside the M x N block"
    again = torch.full_like(buf, 7.0)
    ops.gemm_nt(A, W, again, M, N, K, K, K, ldc, bias=bias)
    assert torch.equal(buf, again), "launch-to-launch difference"


@pytest.mark.parametrize("M,N,K", [(45056, 320, 1280), (45056, 320, 320), (9000, 1600, 640)])
def test_gemm_nt_ragged_rows_on_the_192x320_tile(ops, M, N, K):
    """round 6: N % 320 == 0 with M not a multiple of 192 (pvlt_medium at 384 px: 45056 rows per stage-3 launch = 234.67 row tiles = one round at 92 %) on the ragged
    form of the 8-phase 192 x 320 tile: the plain and the residual epilogue (the K = 320 residual product stays on the 128-wide kernel, as with whole tiles), guard
    rows behind the last one untouched, bit-identical second launch, fp32 reference."""
    from mvlt_amd._lib import last_kernel
    dt = torch.bfloat16
    A, W = rnd(M, K, dtype=dt), rnd(N, K, dtype=dt, seed=1, scale=0.2)
    bias = rnd(N, dtype=torch.float32, seed=2)
    ref = A.float() @ W.float().t() + bias
    for odt in (dt, torch.float32):
        buf = torch.full((M + 192, N), 7.0, device=dev(), dtype=odt)
        ops.gemm_nt(A, W, buf, M, N, K, K, K, N, bias=bias)
        assert "gemm_nt_p8_kernel<1, 3, 3, 2, true>" in last_kernel(), last_kernel()
        assert maxrel(buf[:M].float(), ref) < TOL[dt]
        assert (buf[M:] == 7.0).all(), "wrote behind the last row"
        again = torch.full_like(buf, 7.0)
        ops.gemm_nt(A, W, again, M, N, K, K, K, N, bias=bias)
        assert torch.equal(buf, again), "launch-to-launch difference"
    Bsz = 8
    rps = (M + Bsz - 1) // Bsz
    scale = torch.tensor([0.0, 1.0 / 0.9, 1.0, 1.0 / 0.9, 0.0, 1.0, 1.0, 1.0 / 0.9], device=dev())
    Rres = rnd(M, N, dtype=torch.float32, seed=7)
    want = ref * scale.repeat_interleave(rps)[:M, None] + Rres
    out = torch.full((M + 192, N), 7.0, device=dev())
    out[:M] = Rres
    ops.gemm_nt(A, W, out, M, N, K, K, K, N, bias=bias, row_scale=scale, rows_per_scale=rps, R=out)
    assert ("gemm_nt_p8_kernel<2, 3, 3, 2, true>" in last_kernel()) == (K >= 640), last_kernel()
    assert maxrel(out[:M], want) < TOL[dt] and (out[M:] == 7.0).all()
    o16 = torch.full((M + 192, N), 7.0, device=dev(), dtype=dt)                     # fp32 residual beside a bf16 C (the last block of a stage)
    ops.gemm_nt(A, W, o16, M, N, K, K, K, N, bias=bias, row_scale=scale, rows_per_scale=rps, R=Rres)
    assert torch.equal(o16[:M], out[:M].to(dt)) and (o16[M:] == 7.0).all()


@pytest.mark.parametrize("M,N", [(6400, 2048), (6144, 2048), (7680, 1600)])      # -> 256 x 256, 192 x 256, 192 x 320 tiles (host dispatch by whole rounds)
@pytest.mark.parametrize("K", [128, 320, 512, 64 * 7])
def test_gemm_nt_8phase_tiles(ops, K, M, N):
    """the 8-wave kernels with the 8-phase K-loop (gemm_nt_p8_kernel; taken from 192 tiles on): every epilogue they carry, odd and even
    k-tile counts (the K-loop is unrolled by two k-tiles with a two-k-tile-deep refill pipeline), a result that must not depend on
    the launch (run twice, bit-identical: a refill landing under a late reader shows as a differing tile), and agreement with the
    fp32 reference.  (The 192 x 320 tile carries the plain and the residual epilogue; its other cases run on the 128-wide kernels.)"""
    dt = torch.bfloat16
    Bsz = 5 if M % 5 == 0 and (M // 5) % 8 == 0 else 6
    A, W = rnd(M, K, dtype=dt), rnd(N, K, dtype=dt, seed=1, scale=0.2)
    bias = rnd(N, dtype=torch.float32, seed=2)
    pre = A.float() @ W.float().t()
    ref_pre = pre + bias
    # EPI 1: plain + bias, bf16 and fp32 outputs
    for odt in (dt, torch.float32):
        out = torch.empty(M, N, device=dev(), dtype=odt)
        ops.gemm_nt(A, W, out, M, N, K, K, K, N, bias=bias)
        assert maxrel(out.float(), ref_pre) < TOL[dt], (K, odt)
        out_b = torch.empty_like(out)
        ops.gemm_nt(A, W, out_b, M, N, K, K, K, N, bias=bias)
        assert torch.equal(out, out_b), "launch-to-launch difference"
    # EPI 3: GELU with the pre-activation stored
    out, H = torch.empty(M, N, device=dev(), dtype=dt), torch.empty(M, N, device=dev(), dtype=dt)
    ops.gemm_nt(A, W, out, M, N, K, K, K, N, bias=bias, act=1, H=H)
    assert maxrel(H.float(), ref_pre) < TOL[dt] and maxrel(out.float(), F.gelu(ref_pre)) < TOL[dt]
    # EPI 4: x gelu'(H)
    Hin = rnd(M, N, dtype=dt, seed=5)
    out2 = torch.empty_like(out)
    ops.gemm_nt(A, W, out2, M, N, K, K, K, N, act=2, H=Hin)
    h = Hin.float().requires_grad_(True)
    F.gelu(h).sum().backward()
    assert maxrel(out2.float(), pre * h.grad) < TOL[dt]
    # EPI 2: bias, per-sample factor, fp32 residual in place
    Rres = rnd(M, N, dtype=torch.float32, seed=7)
    scale = torch.tensor([0.0, 1.0 / 0.9, 1.0, 1.0 / 0.9, 0.0, 1.0], device=dev())[:Bsz].contiguous()
    out3 = Rres.clone()
    ops.gemm_nt(A, W, out3, M, N, K, K, K, N, bias=bias, row_scale=scale, rows_per_scale=M // Bsz, R=out3)
    assert maxrel(out3, ref_pre * scale.repeat_interleave(M // Bsz)[:, None] + Rres) < TOL[dt]
    Rb = rnd(M, N, dtype=dt, seed=8)
    out3b = torch.empty_like(Rb)
    ops.gemm_nt(A, W, out3b, M, N, K, K, K, N, R=Rb)
    assert maxrel(out3b.float(), pre + Rb.float()) < TOL[dt]
    # fp32 residual beside a bf16 C (r_fp32: the last block of stages 3-4 hands over the operand copy directly) == the fp32 result rounded once
    out3c = torch.empty(M, N, device=dev(), dtype=dt)
    ops.gemm_nt(A, W, out3c, M, N, K, K, K, N, bias=bias, row_scale=scale, rows_per_scale=M // Bsz, R=Rres)
    assert torch.equal(out3c, out3.to(dt))
    # EPI 5: column sum / sum of squares of the stored values
    cs, cq = torch.zeros(N, device=dev()), torch.zeros(N, device=dev())
    out5 = torch.empty(M, N, device=dev(), dtype=torch.float32)
    ops.gemm_nt(A, W, out5, M, N, K, K, K, N, col_sum=cs, col_sumsq=cq)
    assert maxrel(out5, pre) < TOL[dt] and rel(cs, out5.sum(0)) < 1e-3 and rel(cq, (out5 * out5).sum(0)) < 1e-3
    # batch-strided output rows (a token sub-range of a (B, N_tok, C) buffer)
    rows, stride, off = M // Bsz, M // Bsz + 40, 24
    buf = torch.zeros(Bsz * stride, N, device=dev(), dtype=dt)
    from mvlt_amd._lib import rowmap
    ops.gemm_nt(A, W, buf, M, N, K, K, K, N, bias=bias, c_map=rowmap(rows, stride, off))
    got = buf.view(Bsz, stride, N)[:, off:off + rows].reshape(M, N)
    assert maxrel(got.float(), ref_pre) < TOL[dt]
    assert float(buf.view(Bsz, stride, N)[:, :off].abs().max()) == 0.0 and float(buf.view(Bsz, stride, N)[:, off + rows:].abs().max()) == 0.0


@pytest.mark.parametrize("N,K,M,out_dtype", [(64, 64, 8448, torch.float32), (128, 128, 1000, torch.float32), (64, 64, 200, torch.bfloat16)])
def test_gemm_nt_layernorm_epilogue(ops, N, K, M, out_dtype):
    """attn.proj + DropPath + residual with Block.norm2 of the finished row on the epilogue (reference libs/pvlt.py:140-142):
    C as without the LayerNorm (bit-identical to the plain residual epilogue), post_y = LN(C) of the fp32 row, statistics saved."""
    dt = torch.bfloat16
    Bsz = 8 if M % 8 == 0 else 1
    A, W = rnd(M, K, dtype=dt), rnd(N, K, dtype=dt, seed=1, scale=0.3)
    bias = rnd(N, dtype=torch.float32, seed=2)
    R = rnd(M, N, dtype=out_dtype, seed=3, scale=2.0)
    scale = (torch.arange(Bsz, device=dev()) % 3 != 0).float() / 0.9
    g, b = rnd(N, dtype=torch.float32, seed=4) + 1.0, rnd(N, dtype=torch.float32, seed=5)
    plain = torch.empty(M, N, device=dev(), dtype=out_dtype)
    ops.gemm_nt(A, W, plain, M, N, K, K, K, N, bias=bias, row_scale=scale, rows_per_scale=M // Bsz, R=R)
    out = torch.empty_like(plain)
    y = torch.full((M, N), float("nan"), device=dev(), dtype=dt)
    mean, rstd = torch.empty(M, device=dev()), torch.empty(M, device=dev())
    ops.gemm_nt(A, W, out, M, N, K, K, K, N, bias=bias, row_scale=scale, rows_per_scale=M // Bsz, R=R, post_ln=(g, b, 1e-6, y, mean, rstd))
    assert torch.equal(out, plain)
    x = (A.float() @ W.float().t() + bias) * scale.repeat_interleave(M // Bsz)[:, None] + R.float()      # the fp32 row the kernel normalises
    ref = F.layer_norm(x, (N,), g, b, 1e-6)
    assert torch.isfinite(y.float()).all()
    assert maxrel(y.float(), ref) < TOL[dt]
    assert maxrel(mean, x.mean(1)) < 2e-3 and maxrel(rstd, (x.var(1, unbiased=False) + 1e-6).rsqrt()) < 2e-3


@pytest.mark.parametrize("M,N,K,S", [(1490, 768, 30528, 16), (130, 100, 1000, 4), (64, 64, 64, 8)])
def test_gemm_nt_split_k(ops, M, N, K, S):
    """K cut over S workgroups per tile, fp32 atomics into a zeroed C (input gradient of the tied MLM decoder)."""
    A, W = rnd(M, K, dtype=torch.bfloat16), rnd(N, K, dtype=torch.bfloat16, seed=1, scale=0.05)
    bias = rnd(N, dtype=torch.float32, seed=2)
    out = torch.zeros(M, N, device=dev(), dtype=torch.float32)
    ops.gemm_nt(A, W, out, M, N, K, K, K, N, bias=bias, split_k=S)
    assert maxrel(out, A.float() @ W.float().t() + bias) < TOL[torch.bfloat16]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K", [(1000, 192, 576), (300, 64, 64), (129, 24, 128)])
def test_gemm_nt_column_statistics(ops, dtype, M, N, K):
    """BatchNorm batch statistics on the conv epilogue: per-column sum and sum of squares of the stored fp32 output."""
    A, W = rnd(M, K, dtype=dtype), rnd(N, K, dtype=dtype, seed=1, scale=0.2)
    out = torch.empty(M, N, device=dev(), dtype=torch.float32)
    cs, cq = torch.zeros(N, device=dev()), torch.zeros(N, device=dev())
    ops.gemm_nt(A, W, out, M, N, K, K, K, N, col_sum=cs, col_sumsq=cq)
    assert maxrel(cs, out.sum(0)) < 1e-4
    assert maxrel(cq, (out * out).sum(0)) < 1e-4
    cs4, cq4 = torch.zeros(4, N, device=dev()), torch.zeros(4, N, device=dev())       # interleaved accumulators
    ops.gemm_nt(A, W, out, M, N, K, K, K, N, col_sum=cs4, col_sumsq=cq4, col_copies=4)
    assert maxrel(cs4.sum(0), out.sum(0)) < 1e-4 and maxrel(cq4.sum(0), (out * out).sum(0)) < 1e-4
    if M > 128 * 4:
        assert (cs4.abs().sum(1) > 0).all()
    if dtype == torch.bfloat16 and N % 8 == 0:
        # fp16 output (out_dtype 2: the pre-BatchNorm conv output on the bf16 path): the fp32 result rounded to fp16, statistics of the ROUNDED values
        o16 = torch.empty(M, N, device=dev(), dtype=torch.float16)
        c16, q16 = torch.zeros(N, device=dev()), torch.zeros(N, device=dev())
        ops.gemm_nt(A, W, o16, M, N, K, K, K, N, col_sum=c16, col_sumsq=q16)
        assert torch.equal(o16, out.to(torch.float16))
        assert maxrel(c16, o16.float().sum(0)) < 1e-4 and maxrel(q16, (o16.float() ** 2).sum(0)) < 1e-4


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gemm_nt_token_subrange(ops, dtype):
    """A = text tokens [HW, HW+T) of a (B, N, C) buffer; C written into another buffer's text range."""
    from mvlt_amd._lib import rowmap
    Bsz, HW, T, Cin, Cout = 3, 100, 20, 64, 128
    X = rnd(Bsz, HW + T, Cin, dtype=dtype)
    W = rnd(Cout, Cin, dtype=dtype, seed=1, scale=0.2)
    Y = torch.zeros(Bsz, HW + T + 5, Cout, device=dev(), dtype=dtype)
    ops.gemm_nt(X, W, Y, Bsz * T, Cout, Cin, Cin, Cin, Cout,
                a_map=rowmap(T, HW + T, HW), c_map=rowmap(T, HW + T + 5, HW + 5))
    ref = X[:, HW:].float() @ W.float().t()
    assert maxrel(Y[:, HW + 5:].float(), ref) < TOL[dtype]
    assert Y[:, :HW + 5].abs().max().item() == 0.0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("r,Hin,Cin,Cout,T", [(2, 8, 64, 128, 5), (8, 16, 64, 64, 3), (2, 6, 320, 512, 4), (4, 8, 128, 128, 0)])
def test_gemm_nt_patch_gather_and_scatter(ops, dtype, r, Hin, Cin, Cout, T):
    """kernel==stride conv on token-major data == F.conv2d on the NCHW view; scatter == its input gradient."""
    from mvlt_amd._lib import patchmap
    Bsz = 2
    Win = Hin
    HWi = Hin * Win
    Ho = Hin // r
    X = rnd(Bsz, HWi + T, Cin, dtype=dtype)
    Wc = rnd(Cout, Cin, r, r, dtype=dtype, seed=1, scale=0.1)
    bias = rnd(Cout, dtype=torch.float32, seed=2)
    Wk = Wc.permute(0, 2, 3, 1).reshape(Cout, r * r * Cin).contiguous()       # [out][di][dj][c]
    K = r * r * Cin
    M = Bsz * Ho * Ho
    pm = patchmap(r, Win, HWi + T, Ho * Ho, Ho, Cin)
    out = torch.empty(M, Cout, device=dev(), dtype=dtype)
    ops.gemm_nt(X, Wk, out, M, Cout, K, Cin, K, Cout, a_map=pm, bias=bias)
    img = X[:, :HWi].float().transpose(1, 2).reshape(Bsz, Cin, Hin, Win)
    ref = F.conv2d(img, Wc.float(), bias, stride=r).flatten(2).transpose(1, 2).reshape(M, Cout)
    assert maxrel(out.float(), ref) < TOL[dtype]
    # dgrad: dX(tokens) = scatter(dY @ Wk)   (B operand = Wk^T : [K, Cout])
    dY = rnd(M, Cout, dtype=dtype, seed=3)
    WkT = Wk.t().contiguous()
    dX = torch.zeros(Bsz, HWi + T, Cin, device=dev(), dtype=dtype)
    ops.gemm_nt(dY, WkT, dX, M, K, Cout, Cout, Cout, Cin, c_map=pm)
    imgr = img.clone().requires_grad_(True)
    y = F.conv2d(imgr, Wc.float(), None, stride=r).flatten(2).transpose(1, 2).reshape(M, Cout)
    (y * dY.float()).sum().backward()
    refdx = imgr.grad.reshape(Bsz, Cin, HWi).transpose(1, 2)
    assert maxrel(dX[:, :HWi].float(), refdx) < TOL[dtype]
    if T:
        assert dX[:, HWi:].abs().max().item() == 0.0


# ------------------------------------------------------------------ gemm_tn
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N1,N2", [(1000, 64, 64), (4224 * 3, 512, 64), (700, 64, 512), (333, 320, 1280), (129, 30522, 768), (77, 8, 768)])
def test_gemm_tn(ops, dtype, M, N1, N2):
    lda = (N1 + 7) // 8 * 8          # ragged N1 (vocabulary 30522) lives in rows padded to 16 B
    Ap = rnd(M, lda, dtype=dtype)
    A, B = Ap[:, :N1], rnd(M, N2, dtype=dtype, seed=1)
    out = torch.zeros(N1, N2, device=dev(), dtype=torch.float32)
    cs = torch.zeros(N1, device=dev(), dtype=torch.float32)
    ops.gemm_tn(Ap, B, out, M, N1, N2, lda, N2, N2, colsum=cs)
    ref = A.float().t() @ B.float()
    assert maxrel(out, ref) < TOL[dtype]
    assert maxrel(cs, A.float().sum(0)) < TOL[dtype]
    # accumulate semantics: a second call doubles
    ops.gemm_tn(Ap, B, out, M, N1, N2, lda, N2, N2)
    assert maxrel(out, 2 * ref) < TOL[dtype]


@pytest.mark.parametrize("M,N1,N2", [(49152, 512, 2048), (49152, 2048, 512), (16384, 1024, 1024)])
def test_gemm_tn_partial_tiles_reduce_without_atomics(ops, M, N1, N2):
    """Weight gradients whose output is 16 .. 64 whole 256 x 256 tiles (the stage-4 MLP) with a scratch buffer: the 8-wave / 8-phase TN kernel stores bf16 partial tiles
    and an ordered fold adds them to C (mvlt_gemm_tn_args.partials) -- ACCUMULATING into C like the atomic path, within 5e-3 of the fp32 reference (one bf16 rounding per
    m-split), bias gradient included, bit-identical from launch to launch (the atomic path is not), and the launched kernel is the partial-tile one."""
    from mvlt_amd._lib import last_kernel
    dt = torch.bfloat16
    A, B = rnd(M, N1, dtype=dt, scale=0.5), rnd(M, N2, dtype=dt, seed=1, scale=0.5)
    ref = A.float().t() @ B.float()
    scratch = torch.empty(256 * 65536, device=dev(), dtype=dt)
    outs = []
    for _ in range(2):
        Cw, cs = torch.full((N1, N2), 3.0, device=dev()), torch.zeros(N1, device=dev())
        ops.gemm_tn(A, B, Cw, M, N1, N2, N1, N2, N2, colsum=cs, partials=scratch)
        torch.cuda.synchronize()
        assert "tn_fold_kernel" in last_kernel()
        outs.append((Cw, cs))
    assert maxrel(outs[0][0] - 3.0, ref) < 5e-3
    assert maxrel(outs[0][1], A.float().sum(0)) < 1e-3
    assert torch.equal(outs[0][0], outs[1][0])
    # a scratch that is too small, or none: the atomic kernel, same answer to fp32-atomic accuracy
    Cw = torch.zeros(N1, N2, device=dev())
    ops.gemm_tn(A, B, Cw, M, N1, N2, N1, N2, N2, partials=scratch[:1024])
    torch.cuda.synchronize()
    assert "gemm_tn_dma_kernel" in last_kernel() and maxrel(Cw, ref) < 1e-3


@pytest.mark.parametrize("M,N1,N2,ldc", [(1490, 30522, 768, 768), (1000, 500, 264, 272), (4096, 128, 128, 128)])
def test_gemm_tn_overwrite_stores_instead_of_adding(ops, M, N1, N2, ldc):
    """mvlt_gemm_tn_args.c_overwrite (round 6): C = A^T B by ONE m-split and plain 16-byte stores instead of fp32 atomics -- the vocabulary decoder's weight gradient
    (1490 selected rows x 30522 words x 768, the first writer of its slice of the zeroed gradient buffer).  The buffer is pre-filled with 7 to show that nothing is added to
    what was there, the columns behind N2 and the row behind N1 keep their 7s, the bias gradient (column sums of A) still accumulates, second launch bit-identical."""
    dt = torch.bfloat16
    A, B = rnd(M, N1, dtype=dt, scale=0.5), rnd(M, N2, dtype=dt, seed=1, scale=0.5)
    lda = (N1 + 7) // 8 * 8
    Ap = torch.zeros(M, lda, device=dev(), dtype=dt)
    Ap[:, :N1] = A
    ref = A.float().t() @ B.float()
    outs = []
    for _ in range(2):
        Cw, cs = torch.full((N1 + 1, ldc), 7.0, device=dev()), torch.full((N1,), 1.0, device=dev())
        ops.gemm_tn(Ap, B, Cw, M, N1, N2, lda, N2, ldc, colsum=cs, overwrite=True)
        torch.cuda.synchronize()
        outs.append((Cw, cs))
    assert maxrel(outs[0][0][:N1, :N2], ref) < 1e-3
    assert (outs[0][0][N1] == 7.0).all() and (outs[0][0][:, N2:] == 7.0).all()
    assert maxrel(outs[0][1] - 1.0, A.float().sum(0)) < 1e-3
    assert torch.equal(outs[0][0], outs[1][0])


@pytest.mark.parametrize("M,N1,N2", [(45056, 1280, 320), (45056, 320, 1280), (49152, 1152, 320), (32768, 640, 1600)])
def test_gemm_tn_192x320_tiles_of_the_8phase_loop(ops, M, N1, N2):
    """round 6: one side a multiple of 320, the other >= 1024 (the fc weight gradients of stage 3: dW1 1280 x 320, dW2 320 x 1280) on 192 x 320 tiles of the 8-wave /
    8-phase TN loop -- the last row tile ragged (1280 = 6.67 x 192), the 320-side-first shapes computed with swapped operands and stored transposed -- bf16 partial tiles + the
    ordered fold: accumulates into C within 5e-3 of the fp32 product, bias gradient (column sums of A) included on either side, bit-identical second launch, and the kernel that
    ran is the 192 x 320 instantiation.  (The last case does not qualify -- 640 x 1600: neither side is one-to-two 320-wide column tiles beside >= 1024 rows -- and stays on
    the 128-wide kernel.)"""
    from mvlt_amd._lib import last_kernel
    dt = torch.bfloat16
    A, B = rnd(M, N1, dtype=dt, scale=0.5), rnd(M, N2, dtype=dt, seed=1, scale=0.5)
    ref = A.float().t() @ B.float()
    scratch = torch.empty(256 * 65536, device=dev(), dtype=dt)
    outs = []
    for _ in range(2):
        Cw, cs = torch.full((N1 + 1, N2), 3.0, device=dev()), torch.zeros(N1 + 8, device=dev())
        ops.gemm_tn(A, B, Cw, M, N1, N2, N1, N2, N2, colsum=cs[:N1], partials=scratch, defer_fold=True)
        name = last_kernel()
        ops.tn_fold_flush(scratch)
        torch.cuda.synchronize()
        outs.append((Cw, cs))
    want = "gemm_tn_p8_kernel<3, 3, 2, " + ("true" if N2 % 320 == 0 and N1 >= 1024 else "false")
    if (N1, N2) == (640, 1600):
        assert "gemm_tn_dma_kernel" in name, name
    else:
        assert want in name, name
    assert maxrel(outs[0][0][:N1] - 3.0, ref) < 5e-3
    assert (outs[0][0][N1] == 3.0).all() and (outs[0][1][N1:] == 0).all(), "wrote past the ragged tile"
    assert maxrel(outs[0][1][:N1], A.float().sum(0)) < 1e-3
    assert torch.equal(outs[0][0], outs[1][0])


@pytest.mark.parametrize("M,N1,N2", [(98304, 320, 320), (49152, 512, 512), (40000, 328, 512)])
def test_gemm_tn_partial_tiles_on_the_128_wide_kernel(ops, M, N1, N2):
    """The 128 x 128 weight-gradient kernel in partial-tile mode (many m-splits on a mid-sized output: the q / proj gradients of stages 3-4): accumulates into C within 5e-3 of
    the fp32 product, ragged tiles and a ragged last split included, bit-identical from launch to launch, the bias gradient unchanged."""
    from mvlt_amd._lib import last_kernel
    dt = torch.bfloat16
    A, B = rnd(M, N1, dtype=dt, scale=0.5), rnd(M, N2, dtype=dt, seed=1, scale=0.5)
    ref = A.float().t() @ B.float()
    scratch = torch.empty(256 * 65536, device=dev(), dtype=dt)
    outs = []
    for _ in range(2):
        Cw, cs = torch.full((N1, N2), -2.0, device=dev()), torch.zeros(N1, device=dev())
        ops.gemm_tn(A, B, Cw, M, N1, N2, N1, N2, N2, colsum=cs, partials=scratch)
        torch.cuda.synchronize()
        assert "tn_fold_kernel" in last_kernel()
        outs.append((Cw, cs))
    assert maxrel(outs[0][0] + 2.0, ref) < 5e-3
    assert maxrel(outs[0][1], A.float().sum(0)) < 1e-3
    assert torch.equal(outs[0][0], outs[1][0])


def test_gemm_tn_deferred_folds_equal_immediate_ones(ops):
    """mvlt_gemm_tn_args.defer_fold: the partial tiles of several weight-gradient launches stay in the scratch and ONE launch folds them (mvlt_tn_fold_flush, or the library by
    itself when its 32-entry table or the scratch is full, or when a non-deferring launch needs the scratch).  Bit-identical to folding behind every launch, whatever triggered
    the fold; C untouched until then."""
    from mvlt_amd._lib import last_kernel
    dt = torch.bfloat16
    shapes = [(49152, 512, 512), (24576, 320, 320), (16384, 2048, 512), (40000, 328, 512)]
    ops_in = [(rnd(M, N1, dtype=dt, seed=3 * i, scale=0.5), rnd(M, N2, dtype=dt, seed=3 * i + 1, scale=0.5)) for i, (M, N1, N2) in enumerate(shapes)]
    scratch = torch.empty(2048 * 65536, device=dev(), dtype=dt)

    def run(defer, reps=1, small=None):
        outs = []
        for r in range(reps):
            for (M, N1, N2), (A, B) in zip(shapes, ops_in):
                Cw = torch.full((N1, N2), 0.25, device=dev())
                ops.gemm_tn(A, B, Cw, M, N1, N2, N1, N2, N2, partials=scratch if small is None else small, defer_fold=defer)
                outs.append(Cw)
        return outs

    want = run(False)
    torch.cuda.synchronize()
    got = run(True)
    torch.cuda.synchronize()
    assert all(bool((g == 0.25).all()) for g in got)                      # nothing folded yet
    ops.tn_fold_flush()
    torch.cuda.synchronize()
    assert "tn_fold_multi_kernel" in last_kernel()
    for g, w in zip(got, want):
        assert torch.equal(g, w)
    ops.tn_fold_flush()                                                   # nothing pending: no launch, no error
    # two deferring launches into the SAME gradient (the kv weights take their text rows and their image rows from two GEMMs): folded one after the other, not in one launch
    (A0, B0), (A1, B1) = (rnd(49152, 512, dtype=dt, seed=40, scale=0.5), rnd(49152, 512, dtype=dt, seed=41, scale=0.5)), ops_in[0]
    both = []
    for defer in (False, True):
        Cw = torch.zeros(512, 512, device=dev())
        ops.gemm_tn(A0, B0, Cw, 49152, 512, 512, 512, 512, 512, partials=scratch, defer_fold=defer)
        ops.gemm_tn(A1, B1, Cw[:256], 49152, 256, 512, 512, 512, 512, partials=scratch, defer_fold=defer)      # (an overlapping row range of it)
        ops.tn_fold_flush()
        torch.cuda.synchronize()
        both.append(Cw)
    assert torch.equal(both[0], both[1])
    # more launches than the table (32) or the scratch holds: the library folds the earlier ones by itself, the last one is still pending
    got = run(True, reps=5)
    torch.cuda.synchronize()
    assert torch.equal(got[0], want[0]) and torch.equal(got[3], want[3]) and bool((got[-1] == 0.25).all())
    # a non-deferring launch that needs the scratch folds what is pending first
    Cn = torch.zeros(512, 512, device=dev())
    ops.gemm_tn(ops_in[0][0], ops_in[0][1], Cn, 49152, 512, 512, 512, 512, 512, partials=scratch)
    torch.cuda.synchronize()
    assert all(torch.equal(g, w) for g, w in zip(got, want * 5)) and torch.equal(Cn + 0.25, want[0])
    # an abandoned pass: pending folds are dropped, nothing is written
    got = run(True)
    ops.tn_fold_discard()
    ops.tn_fold_flush()
    torch.cuda.synchronize()
    assert all(bool((g == 0.25).all()) for g in got)
    # ---- round 6: the pending table is kept PER SCRATCH.  Two owners (two parameter stores in one backward pass): discarding / flushing one leaves the
    # other's entries alone; a launch on another stream folds what its scratch has pending on the OLD stream and orders itself behind that fold.
    other = torch.empty(64 * 65536 * 8, device=dev(), dtype=dt)
    (M, N1, N2), (A, B) = shapes[0], ops_in[0]
    Ca, Cb = torch.full((N1, N2), 0.25, device=dev()), torch.full((N1, N2), 0.25, device=dev())
    ops.gemm_tn(A, B, Ca, M, N1, N2, N1, N2, N2, partials=scratch, defer_fold=True)
    ops.gemm_tn(A, B, Cb, M, N1, N2, N1, N2, N2, partials=other, defer_fold=True)
    ops.tn_fold_discard(other)                                            # the second store starts a new pass: ITS entries go
    ops.tn_fold_flush(other)
    torch.cuda.synchronize()
    assert bool((Ca == 0.25).all()) and bool((Cb == 0.25).all())          # the first store's fold is still pending, untouched
    ops.tn_fold_flush(scratch)
    torch.cuda.synchronize()
    assert torch.equal(Ca, want[0]) and bool((Cb == 0.25).all())
    side = torch.cuda.Stream()
    Ca.fill_(0.25)
    Cs = torch.full((N1, N2), 0.25, device=dev())
    ops.gemm_tn(A, B, Ca, M, N1, N2, N1, N2, N2, partials=scratch, defer_fold=True)            # pending on the current stream
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.gemm_tn(A, B, Cs, M, N1, N2, N1, N2, N2, partials=scratch, defer_fold=True)        # region 0 again: folds Ca on the old stream first, waits for it
        ops.tn_fold_flush(scratch)
    torch.cuda.synchronize()
    assert torch.equal(Ca, want[0]) and torch.equal(Cs, want[0])
    with torch.cuda.stream(side):
        Cs.fill_(0.25)
        ops.gemm_tn(A, B, Cs, M, N1, N2, N1, N2, N2, partials=scratch, defer_fold=True)
    ops.tn_fold_flush(scratch)                                            # a reader on ANOTHER stream than the producers': ordered behind the fold by an event
    got_now = Cs.clone()
    torch.cuda.synchronize()
    assert torch.equal(got_now, want[0])
    # a scratch that holds two of the partial sets but not three: folded when the next one does not fit
    small = scratch[: 48 * 512 * 512 * 2 + 4096]
    got = run(True, small=small)
    ops.tn_fold_flush()
    torch.cuda.synchronize()
    for g, w in zip(got, want):
        assert maxrel(g - 0.25, w - 0.25) < 5e-3                          # (the launches that did not fit took the atomic path)


def test_gemm_tn_fused_input_gradient_is_never_skipped_silently(ops):
    """every launch path of mvlt_gemm_tn that cannot produce dgrad_out must refuse -- a silently unwritten input gradient is a wrong
    gradient.  fp32 operands (the generic kernel) with dgrad_out raise; a row-strided dX view (a column slice of a wider buffer) keeps its
    neighbours intact because its own row pitch reaches the kernel."""
    import ctypes as C_
    from mvlt_amd import _lib as L
    C, M = 64, 640
    dY, X = rnd(M, C, dtype=torch.float32), rnd(M, C, dtype=torch.float32, seed=1)
    WT = rnd(C, C, dtype=torch.bfloat16, seed=2)
    dW = torch.zeros(C, C, device=dev())
    dX = torch.zeros(M, C, device=dev(), dtype=torch.bfloat16)
    a = L.GemmTNArgs(L.ptr(dY), L.ptr(X), L.ptr(dW), M, C, C, C, C, C, 1, L.rowmap(), L.rowmap(), None, 0, None, 0, 0, 0, L.ptr(WT), L.ptr(dX), C)
    assert L.lib.mvlt_gemm_tn(C_.byref(a), L.stream_ptr()) != 0 and b"dgrad_out" in L.lib.mvlt_last_error()
    dt = torch.bfloat16
    dYb, Xb = dY.to(dt), X.to(dt)
    wide = torch.full((M, 2 * C), 7.0, device=dev(), dtype=dt)
    ops.gemm_tn(dYb, Xb, dW, M, C, C, C, C, C, dgrad=(WT, wide[:, :C]))
    torch.cuda.synchronize()
    assert (wide[:, C:] == 7.0).all()                                     # the other half of every row is untouched
    assert maxrel(wide[:, :C].float(), dYb.float() @ WT.float().t()) < TOL[dt]


@pytest.mark.parametrize("C,M", [(64, 64 * 300 + 17), (128, 64 * 150 + 40), (64, 130), (128, 4224 * 8)])
def test_gemm_tn_with_fused_input_gradient(ops, C, M):
    """weight gradient + bias gradient + INPUT gradient of a C x C Linear from one pass over dY (mvlt_gemm_tn dgrad_*; the q / proj
    projections of stages 1-2): dW += dY^T X, db += colsum(dY), dX = dY W -- against fp32 references, ragged last tile included, and
    the two-launch form (gemm_tn + gemm_nt) as a second witness for dX."""
    dt = torch.bfloat16
    dY, X = rnd(M, C, dtype=dt, scale=0.5), rnd(M, C, dtype=dt, seed=1, scale=0.5)
    W = rnd(C, C, dtype=dt, seed=2, scale=C ** -0.5)                  # [out][in]
    WT = W.t().contiguous()
    dW, db = torch.zeros(C, C, device=dev()), torch.zeros(C, device=dev())
    dX = torch.full((M, C), float("nan"), device=dev(), dtype=dt)
    ops.gemm_tn(dY, X, dW, M, C, C, C, C, C, colsum=db, dgrad=(WT, dX))
    assert maxrel(dW, dY.float().t() @ X.float()) < 1e-3
    assert maxrel(db, dY.float().sum(0)) < 1e-3
    assert torch.isfinite(dX.float()).all()
    assert maxrel(dX.float(), dY.float() @ W.float()) < TOL[dt]
    dX2 = torch.empty_like(dX)
    ops.gemm_nt(dY, WT, dX2, M, C, C, C, C, C)
    assert maxrel(dX.float(), dX2.float()) < 1e-2


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gemm_tn_patch_gather(ops, dtype):
    from mvlt_amd._lib import patchmap
    Bsz, r, Hin, Cin, Cout, T = 2, 2, 8, 64, 128, 6
    HWi, Ho = Hin * Hin, Hin // r
    X = rnd(Bsz, HWi + T, Cin, dtype=dtype)
    M, K = Bsz * Ho * Ho, r * r * Cin
    dY = rnd(M, Cout, dtype=dtype, seed=3)
    dW = torch.zeros(Cout, K, device=dev(), dtype=torch.float32)
    ops.gemm_tn(dY, X, dW, M, Cout, K, Cout, Cin, K, b_map=patchmap(r, Hin, HWi + T, Ho * Ho, Ho, Cin))
    img = X[:, :HWi].float().transpose(1, 2).reshape(Bsz, Cin, Hin, Hin)
    Wc = torch.zeros(Cout, Cin, r, r, device=dev(), requires_grad=True)
    y = F.conv2d(img, Wc, None, stride=r).flatten(2).transpose(1, 2).reshape(M, Cout)
    (y * dY.float()).sum().backward()
    ref = Wc.grad.permute(0, 2, 3, 1).reshape(Cout, K)
    assert maxrel(dW, ref) < TOL[dtype]


# ------------------------------------------------------------------ layernorm
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,Cdim,eps", [(1000, 64, 1e-6), (517, 128, 1e-5), (300, 320, 1e-5), (64, 512, 1e-6), (130, 768, 1e-12)])
def test_layernorm_fwd_bwd(ops, dtype, rows, Cdim, eps):
    x = rnd(rows, Cdim, dtype=dtype, scale=2.0) + 0.5
    g = (1 + 0.2 * rnd(Cdim, dtype=torch.float32, seed=1))
    b = 0.1 * rnd(Cdim, dtype=torch.float32, seed=2)
    y = torch.empty_like(x)
    mean = torch.empty(rows, device=dev())
    rstd = torch.empty(rows, device=dev())
    ops.layernorm_fwd(x, y, g, b, rows, Cdim, Cdim, Cdim, eps, mean=mean, rstd=rstd)
    xr = x.float().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (Cdim,), gr, br, eps)
    assert maxrel(y.float(), ref) < TOL[dtype]
    dy = rnd(rows, Cdim, dtype=dtype, seed=4)
    ref.backward(dy.float())
    dx = rnd(rows, Cdim, dtype=dtype, seed=5)       # pre-existing gradient (accumulate mode)
    dx0 = dx.clone()
    dg = torch.zeros(Cdim, device=dev())
    db = torch.zeros(Cdim, device=dev())
    ops.layernorm_bwd(dy, x, dx, g, mean, rstd, rows, Cdim, Cdim, Cdim, Cdim, dgamma=dg, dbeta=db, accumulate=True)
    assert maxrel(dx.float() - dx0.float(), xr.grad) < (2 * TOL[dtype])
    assert maxrel(dg, gr.grad) < TOL[dtype]
    assert maxrel(db, br.grad) < TOL[dtype]
    dx2 = torch.empty_like(x)
    ops.layernorm_bwd(dy, x, dx2, g, mean, rstd, rows, Cdim, Cdim, Cdim, Cdim)
    assert maxrel(dx2.float(), xr.grad) < TOL[dtype]


@pytest.mark.parametrize("Cdim", [64, 128, 320, 512])
def test_layernorm_chained_second_norm(ops, Cdim):
    """mvlt_layernorm_fwd with y2: the embedding LayerNorm (+ pos-embed, rows mapped into the token buffer) and the first block's norm1
    of the row it just wrote, in one pass -- against two separate launches"""
    bf = torch.bfloat16
    B, HW, T = 3, 20, 7
    N = HW + T
    pre = rnd(B * HW, Cdim, dtype=bf)
    g1, b1 = 1 + 0.2 * rnd(Cdim, dtype=torch.float32, seed=1), 0.1 * rnd(Cdim, dtype=torch.float32, seed=2)
    g2, b2 = 1 + 0.2 * rnd(Cdim, dtype=torch.float32, seed=3), 0.1 * rnd(Cdim, dtype=torch.float32, seed=4)
    pos = rnd(HW, Cdim, dtype=torch.float32, seed=5)
    from mvlt_amd._lib import rowmap
    ymap = rowmap(HW, N, 0)
    x_ref = torch.zeros(B, N, Cdim, device=dev())
    ops.layernorm_fwd(pre, x_ref, g1, b1, B * HW, Cdim, Cdim, Cdim, 1e-5, add=pos, add_rows=HW, y_map=ymap)
    y2_ref = torch.zeros(B, N, Cdim, device=dev(), dtype=bf)
    m_ref, r_ref = torch.zeros(B * N, device=dev()), torch.zeros(B * N, device=dev())
    ops.layernorm_fwd(x_ref, y2_ref, g2, b2, B * N, Cdim, Cdim, Cdim, 1e-6, mean=m_ref, rstd=r_ref)
    x = torch.zeros(B, N, Cdim, device=dev())
    y2 = torch.zeros(B, N, Cdim, device=dev(), dtype=bf)
    m2, r2 = torch.zeros(B * N, device=dev()), torch.zeros(B * N, device=dev())
    mean, rstd = torch.empty(B * HW, device=dev()), torch.empty(B * HW, device=dev())
    ops.layernorm_fwd(pre, x, g1, b1, B * HW, Cdim, Cdim, Cdim, 1e-5, mean=mean, rstd=rstd, add=pos, add_rows=HW, y_map=ymap,
                      chain=(g2, b2, 1e-6, y2, m2, r2))
    assert torch.equal(x, x_ref)
    img = torch.zeros(B, N, dtype=torch.bool, device=dev())
    img[:, :HW] = True
    sel = img.view(-1)
    assert maxrel(m2[sel], m_ref[sel]) < 1e-5 and maxrel(r2[sel], r_ref[sel]) < 1e-5
    assert (m2[~sel] == 0).all() and (y2[:, HW:] == 0).all()                       # text rows belong to the other launch
    d = (y2[:, :HW].float() - y2_ref[:, :HW].float()).abs().max().item()
    assert d <= 2 ** -7 * y2_ref.float().abs().max().item()                        # at most one bf16 ulp apart


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_layernorm_into_concat_with_pos(ops, dtype):
    """LN + pos-embed add + write into the image-token range of the concatenated (B, HW+T, C) buffer."""
    from mvlt_amd._lib import rowmap
    Bsz, HW, T, Cdim = 3, 50, 7, 128
    x = rnd(Bsz * HW, Cdim, dtype=dtype)
    g, b = 1 + 0.1 * rnd(Cdim, dtype=torch.float32, seed=1), 0.1 * rnd(Cdim, dtype=torch.float32, seed=2)
    pos = rnd(HW, Cdim, dtype=torch.float32, seed=3)
    out = torch.zeros(Bsz, HW + T, Cdim, device=dev(), dtype=dtype)
    ops.layernorm_fwd(x, out, g, b, Bsz * HW, Cdim, Cdim, Cdim, 1e-5, add=pos, add_rows=HW, y_map=rowmap(HW, HW + T, 0))
    ref = F.layer_norm(x.float(), (Cdim,), g, b, 1e-5).reshape(Bsz, HW, Cdim) + pos
    assert maxrel(out[:, :HW].float(), ref) < TOL[dtype]
    assert out[:, HW:].abs().max().item() == 0
    s = torch.empty(HW + T, Cdim, device=dev())
    ops.batch_sum(out, s, Bsz, HW + T, Cdim, HW + T, Cdim)
    assert maxrel(s, out.float().sum(0)) < 1e-5
    # the rows from `split` on ADDED to a second destination (the trunk's text_pos_embed gradient), the others stored as before
    full = rnd(Bsz, HW + T, Cdim, dtype=dtype, seed=11)
    s2 = torch.full((HW + T, Cdim), 7.0, device=dev())
    acc = torch.full((T, Cdim), 0.5, device=dev())
    ops.batch_sum(full, s2, Bsz, HW + T, Cdim, HW + T, Cdim, acc2=acc, split=HW)
    want = full.float().sum(0)
    assert maxrel(s2[:HW], want[:HW]) < 1e-5 and (s2[HW:] == 7.0).all() and maxrel(acc, want[HW:] + 0.5) < 1e-5


def test_head_grad_prep(ops):
    """backward prologue of the small classification heads: padded operand copy of dlogits + column sums added to both bias gradients"""
    for B, n in ((256, 2), (64, 48), (37, 122)):
        n_pad = (n + 7) // 8 * 8
        dlog = rnd(B, n, dtype=torch.float32, seed=n)
        for dt in (torch.bfloat16, torch.float32):
            dl = torch.full((B, n_pad), 9.0, device=dev(), dtype=dt)
            b1, b2 = torch.full((n,), 1.0, device=dev()), torch.full((n,), 2.0, device=dev())
            ops.head_grad_prep(dlog, dl, b1, b2)
            assert torch.equal(dl[:, :n], dlog.to(dt)) and (dl[:, n:] == 0).all()
            assert maxrel(b1 - 1.0, dlog.sum(0)) < 1e-5 and maxrel(b2 - 2.0, dlog.sum(0)) < 1e-5


# ------------------------------------------------------------------ attention
def attn_ref(q, kv, H, scale):
    B, N, Cdim = q.shape
    M = kv.shape[1]
    hd = Cdim // H
    qh = q.float().reshape(B, N, H, hd).permute(0, 2, 1, 3)
    k = kv.float()[..., :Cdim].reshape(B, M, H, hd).permute(0, 2, 1, 3)
    v = kv.float()[..., Cdim:].reshape(B, M, H, hd).permute(0, 2, 1, 3)
    s = (qh @ k.transpose(-1, -2)) * scale
    o = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, N, Cdim)
    return o, torch.logsumexp(s, dim=-1)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,H,N,M", [(2, 1, 4224, 192), (2, 2, 1152, 192), (1, 5, 384, 192), (3, 8, 192, 192),
                                     (1, 1, 9344, 272), (2, 8, 272, 272), (2, 2, 29, 29), (1, 1, 596, 29), (1, 3, 100, 320),
                                     # round-3 forward (<= 192 keys): padded keys inside the second block, an odd number of key tiles
                                     # (3 + 2), two tiles (1 + 1), and a query count that leaves partial tiles in several chunks
                                     (2, 2, 700, 100), (1, 3, 333, 150), (2, 1, 1500, 64), (1, 2, 260, 40)])
def test_sr_attention_fwd(ops, dtype, B, H, N, M):
    if dtype == torch.float32 and M > 288:
        pytest.skip("fp32 K/V^T of 320 keys exceed 160 KB LDS")
    Cdim = 64 * H
    q = rnd(B, N, Cdim, dtype=dtype)
    kv = rnd(B, M, 2 * Cdim, dtype=dtype, seed=1)
    o = torch.empty_like(q)
    lse = torch.empty(B, H, N, device=dev())
    ops.sr_attention_fwd(q, kv, o, lse, B, H, N, M, Cdim, 2 * Cdim, Cdim, 0, Cdim, 0.125)
    ref, ref_lse = attn_ref(q, kv, H, 0.125)
    assert maxrel(o.float(), ref) < TOL[dtype]
    assert (lse - ref_lse).abs().max().item() < (2e-2 if dtype == torch.bfloat16 else 1e-3)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,H,N,M", [(2, 1, 4224, 192), (2, 2, 1152, 192), (3, 8, 192, 192), (1, 1, 2000, 272), (2, 2, 29, 29), (1, 5, 300, 150)])
def test_sr_attention_bwd(ops, dtype, B, H, N, M):
    Cdim = 64 * H
    q = rnd(B, N, Cdim, dtype=dtype)
    kv = rnd(B, M, 2 * Cdim, dtype=dtype, seed=1)
    do = rnd(B, N, Cdim, dtype=dtype, seed=2)
    o = torch.empty_like(q)
    lse = torch.empty(B, H, N, device=dev())
    ops.sr_attention_fwd(q, kv, o, lse, B, H, N, M, Cdim, 2 * Cdim, Cdim, 0, Cdim, 0.125)
    dq = torch.empty_like(q)
    dkv = torch.zeros(B, M, 2 * Cdim, device=dev(), dtype=torch.float32)
    ops.sr_attention_bwd(q, kv, o, do, lse, dq, dkv, B, H, N, M, Cdim, 2 * Cdim, Cdim, 2 * Cdim, 0, Cdim, 0.125)
    qr, kvr = q.float().requires_grad_(True), kv.float().requires_grad_(True)
    ref, _ = attn_ref(qr, kvr, H, 0.125)
    ref.backward(do.float())
    tol = 3e-2 if dtype == torch.bfloat16 else 2e-3
    assert maxrel(dq.float(), qr.grad) < tol
    assert maxrel(dkv, kvr.grad) < tol


def test_sr_attention_fwd_late_maximum(ops):
    """The round-3 forward takes the FIRST key block's row maximum as the exponent reference and rescales only when the second block tops it
    by more than 2^24.  Random data never takes that branch (cdna guide rule 26: a rare data-dependent branch needs an input that forces
    it): here one key of the second block is aligned with every query so that its score exceeds everything in the first block by far more
    than the threshold -- for half of the queries only (the branch is wave-uniform, the factor per lane), and once within the threshold."""
    bf = torch.bfloat16
    B, H, N, M = 2, 1, 320, 192
    for boost in (60.0, 3.0):                              # 60 / 0.125-scaled: ~2^80 above block 0 (rescale); 3: inside the threshold
        q = rnd(B, N, 64, dtype=bf)
        kv = rnd(B, M, 128, dtype=bf, seed=1)
        kv[:, 150, :64] = 0
        kv[:, 150, :8] = boost                             # key 150 (second block): score = boost * sum(q[:8]) * scale
        q[:, ::2, :8] = q[:, ::2, :8].abs() + 2.0          # every other query: large positive score on that key
        o = torch.empty_like(q)
        lse = torch.empty(B, H, N, device=dev())
        ops.sr_attention_fwd(q, kv, o, lse, B, H, N, M, 64, 128, 64, 0, 64, 0.125)
        ref, ref_lse = attn_ref(q, kv, H, 0.125)
        assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all()
        assert maxrel(o.float(), ref) < TOL[bf], boost
        assert ((lse - ref_lse).abs() / ref_lse.abs().clamp_min(1.0)).max().item() < 2e-2, boost


@pytest.mark.parametrize("B,H,N,M", [(128, 4, 96, 80), (64, 5, 704, 272), (2, 1, 4224, 192)])
def test_sr_attention_bwd_bf16_dkv(ops, B, H, N, M):
    """a bf16 dKV forces one query chunk per (batch, head): dK/dV stored once and directly in bf16 (B * heads >= 512 at 256 px; pvlt_medium's stage 3 at batch 64,
    where mvlt_sr_attention_bwd_chunks says 1 by itself -- two rounds of whole-query workgroups beat three rounds + 22 M atomics; the last case would take two chunks
    with an fp32 dKV and still has to be right with one)."""
    assert ops.sr_attention_bwd_chunks(128, 4, 96, 80, torch.bfloat16) == 1 and ops.sr_attention_bwd_chunks(64, 5, 704, 272, torch.bfloat16) == 1
    assert ops.sr_attention_bwd_chunks(256, 1, 4224, 192, torch.bfloat16) == 2 and ops.sr_attention_bwd_chunks(64, 2, 2432, 272, torch.bfloat16) == 2
    Cdim = 64 * H
    dtype = torch.bfloat16
    q = rnd(B, N, Cdim, dtype=dtype)
    kv = rnd(B, M, 2 * Cdim, dtype=dtype, seed=1)
    do = rnd(B, N, Cdim, dtype=dtype, seed=2)
    o = torch.empty_like(q)
    lse = torch.empty(B, H, N, device=dev())
    ops.sr_attention_fwd(q, kv, o, lse, B, H, N, M, Cdim, 2 * Cdim, Cdim, 0, Cdim, 0.125)
    dq = torch.empty_like(q)
    dkv = torch.full((B, M, 2 * Cdim), float("nan"), device=dev(), dtype=dtype)       # every element must be written
    ops.sr_attention_bwd(q, kv, o, do, lse, dq, dkv, B, H, N, M, Cdim, 2 * Cdim, Cdim, 2 * Cdim, 0, Cdim, 0.125)
    qr, kvr = q.float().requires_grad_(True), kv.float().requires_grad_(True)
    ref, _ = attn_ref(qr, kvr, H, 0.125)
    ref.backward(do.float())
    assert maxrel(dq.float(), qr.grad) < 3e-2
    assert maxrel(dkv.float(), kvr.grad) < 3e-2


# ------------------------------------------------------------------ mixed-dtype LayerNorm (fp32 residual stream, bf16 operands)
def test_sr_attention_bwd_repeatable(ops):
    """Regression: several query tiles per workgroup (stage-1 shape at small batch).  The deferred dQ store of a tile read its
    LDS tile after a barrier that hipcc had emitted without draining the writing wave's own ds_writes: 2-4 stale rows in about
    one launch of 100.  200 launches must agree with the first one."""
    B, H, N, M = 4, 1, 4224, 192
    Cdim = 64 * H
    q = rnd(B, N, Cdim, dtype=torch.bfloat16)
    kv = rnd(B, M, 2 * Cdim, dtype=torch.bfloat16, seed=1)
    do = rnd(B, N, Cdim, dtype=torch.bfloat16, seed=2)
    o = torch.empty_like(q)
    lse = torch.empty(B, H, N, device=dev())
    ops.sr_attention_fwd(q, kv, o, lse, B, H, N, M, Cdim, 2 * Cdim, Cdim, 0, Cdim, 0.125)
    ref = None
    for _ in range(200):
        dq = torch.empty_like(q)
        dkv = torch.zeros(B, M, 2 * Cdim, device=dev(), dtype=torch.float32)
        ops.sr_attention_bwd(q, kv, o, do, lse, dq, dkv, B, H, N, M, Cdim, 2 * Cdim, Cdim, 2 * Cdim, 0, Cdim, 0.125)
        if ref is None:
            ref = (dq.float().clone(), dkv.clone())
            continue
        assert torch.equal(dq.float(), ref[0])                       # dQ has no atomics: bit-identical
        assert maxrel(dkv, ref[1]) < 1e-4                            # dK / dV: fp32 atomics across query chunks


def test_layernorm_bwd_scaled_copy(ops):
    """dx2 = (dx after accumulation) * scale[sample]: the DropPath-scaled gradient written by the same kernel."""
    B, N, Cd = 3, 50, 128
    rows = B * N
    for dtype in (torch.bfloat16, torch.float32):
        x = rnd(rows, Cd, dtype=torch.float32)
        dy = rnd(rows, Cd, dtype=dtype, seed=1)
        gamma = rnd(Cd, dtype=torch.float32, seed=2) + 1.0
        mean, var = x.mean(1), x.var(1, unbiased=False)
        rstd = (var + 1e-6).rsqrt()
        base = rnd(rows, Cd, dtype=dtype, seed=3)
        dx = base.clone()
        dx2 = torch.empty_like(dx)
        sc = torch.tensor([0.0, 1.25, 1.0], device=dev())
        dg, db = torch.zeros(Cd, device=dev()), torch.zeros(Cd, device=dev())
        ops.layernorm_bwd(dy, x, dx, gamma, mean, rstd, rows, Cd, Cd, Cd, Cd, dgamma=dg, dbeta=db, accumulate=True,
                          dx2=dx2, dx2_scale=sc, dx2_rows_per_scale=N, lddx2=Cd)
        xr = x.clone().requires_grad_(True)
        F.layer_norm(xr, (Cd,), gamma, torch.zeros_like(gamma), 1e-6).backward(dy.float())
        want = base.float() + xr.grad
        assert maxrel(dx.float(), want) < TOL[dtype]
        assert maxrel(dx2.float(), want * sc.repeat_interleave(N)[:, None]) < TOL[dtype]
        assert dx2[:N].abs().max().item() == 0


def test_layernorm_mixed_dtypes(ops):
    rows, Cdim, eps = 777, 320, 1e-6
    x = rnd(rows, Cdim, dtype=torch.float32, scale=2.0) + 0.5
    g, b = 1 + 0.2 * rnd(Cdim, dtype=torch.float32, seed=1), 0.1 * rnd(Cdim, dtype=torch.float32, seed=2)
    y = torch.empty(rows, Cdim, device=dev(), dtype=torch.bfloat16)
    mean, rstd = torch.empty(rows, device=dev()), torch.empty(rows, device=dev())
    ops.layernorm_fwd(x, y, g, b, rows, Cdim, Cdim, Cdim, eps, mean=mean, rstd=rstd)
    xr = x.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (Cdim,), g, b, eps)
    assert maxrel(y.float(), ref) < 1e-2
    dy = rnd(rows, Cdim, dtype=torch.bfloat16, seed=4)
    ref.backward(dy.float())
    dx = torch.zeros(rows, Cdim, device=dev(), dtype=torch.bfloat16)
    ops.layernorm_bwd(dy, x, dx, g, mean, rstd, rows, Cdim, Cdim, Cdim, Cdim, accumulate=True)
    assert maxrel(dx.float(), xr.grad) < 1e-2
    # bf16 input -> fp32 output (patch-embed LN writing into the fp32 token buffer)
    xb = x.to(torch.bfloat16)
    y32 = torch.empty(rows, Cdim, device=dev(), dtype=torch.float32)
    ops.layernorm_fwd(xb, y32, g, b, rows, Cdim, Cdim, Cdim, 1e-5)
    assert maxrel(y32, F.layer_norm(xb.float(), (Cdim,), g, b, 1e-5)) < 1e-5


# ------------------------------------------------------------------ BERT embeddings
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_bert_embed_fwd_bwd(ops, dtype):
    B, T, Hd, V = 3, 20, 768, 1000
    ids = torch.randint(1, V, (B, T), device=dev())
    ids[:, -5:] = 0                                           # PAD tail: no gradient into row 0 (padding_idx)
    word = rnd(V, Hd, dtype=torch.float32, scale=0.1)
    pos = rnd(512, Hd, dtype=torch.float32, seed=1, scale=0.1)
    typ = rnd(2, Hd, dtype=torch.float32, seed=2, scale=0.1)
    g, b = 1 + 0.2 * rnd(Hd, dtype=torch.float32, seed=3), 0.1 * rnd(Hd, dtype=torch.float32, seed=4)
    keep = (torch.rand(B * T, Hd, device=dev()) >= 0.1).to(torch.uint8)
    y = torch.empty(B * T, Hd, device=dev(), dtype=dtype)
    mean, rstd = torch.empty(B * T, device=dev()), torch.empty(B * T, device=dev())
    ops.bert_embed_fwd(ids, word, pos, typ, g, b, keep, 0.1, y, mean, rstd, B * T, T, 1e-12)
    wr, pr, tr = word.clone().requires_grad_(True), pos.clone().requires_grad_(True), typ.clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    e = F.embedding(ids, wr, padding_idx=0) + tr[0] + pr[:T]
    ref = F.layer_norm(e, (Hd,), gr, br, 1e-12).reshape(B * T, Hd) * keep.float() / 0.9
    assert maxrel(y.float(), ref) < TOL[dtype]
    dy = rnd(B * T, Hd, dtype=dtype, seed=9)
    ref.backward(dy.float())
    dw, dp, dtp = torch.zeros_like(word), torch.zeros_like(pos), torch.zeros_like(typ)
    dg, db = torch.zeros_like(g), torch.zeros_like(b)
    ops.bert_embed_bwd(dy, ids, word, pos, typ, g, keep, 0.1, mean, rstd, dw, dp, dtp[0], dg, db, B * T, T)
    for a, r in ((dw, wr.grad), (dp, pr.grad), (dtp, tr.grad), (dg, gr.grad), (db, br.grad)):
        assert maxrel(a, r) < 2e-3
    assert dw[0].abs().max().item() == 0.0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_patchify_matches_conv(ops, dtype):
    B, S, k, Cout = 2, 32, 4, 64
    img = torch.rand(B, 3, S, S, device=dev())
    P = torch.empty(B * (S // k) ** 2, 3 * k * k, device=dev(), dtype=dtype)
    ops.patchify(img, P, B, 3, S, S, k)
    W = rnd(Cout, 3, k, k, dtype=torch.float32, scale=0.2)
    ref = F.conv2d(img, W, stride=k).flatten(2).transpose(1, 2).reshape(-1, Cout)
    assert maxrel(P.float() @ W.reshape(Cout, -1).t(), ref) < TOL[dtype]


def test_gemm_nt_split_k_with_patch_gather(ops):
    """split_k on a gathered A operand (the spatial-reduction conv of stage 1: 8 x 8 patches of 64 channels, K = 4096, only 128 output tiles): the K ranges of the splits start
    inside later patch segments; partial sums meet in a zeroed fp32 C, bias added once -- against F.conv2d, and against the unsplit launch"""
    from mvlt_amd._lib import patchmap
    Bsz, side, Cc, r, T = 4, 64, 64, 8, 16
    N = side * side + T
    x = rnd(Bsz, N, Cc, dtype=torch.bfloat16)
    Wc = rnd(Cc, Cc, r, r, dtype=torch.float32, scale=0.05, seed=2)
    bias = rnd(Cc, dtype=torch.float32, seed=3)
    Wk = Wc.permute(0, 2, 3, 1).reshape(Cc, r * r * Cc).to(torch.bfloat16).contiguous()          # [out][kh][kw][cin]
    sr = side // r
    HWr = sr * sr
    pm = patchmap(r, side, N, HWr, sr, Cc)
    img = x[:, : side * side].float().reshape(Bsz, side, side, Cc).permute(0, 3, 1, 2)
    ref = F.conv2d(img, Wk.float().view(Cc, r, r, Cc).permute(0, 3, 1, 2), bias, stride=r).permute(0, 2, 3, 1).reshape(Bsz * HWr, Cc)
    one = torch.empty(Bsz * HWr, Cc, device=dev())
    ops.gemm_nt(x, Wk, one, Bsz * HWr, Cc, r * r * Cc, Cc, r * r * Cc, Cc, a_map=pm, bias=bias)
    for sk in (2, 4, 7):
        out = torch.zeros(Bsz * HWr, Cc, device=dev())
        ops.gemm_nt(x, Wk, out, Bsz * HWr, Cc, r * r * Cc, Cc, r * r * Cc, Cc, a_map=pm, bias=bias, split_k=sk)
        assert maxrel(out, ref) < TOL[torch.bfloat16] and maxrel(out, one) < 1e-5, sk


def test_masked_select_bit_exact(ops):
    for n, frac in ((128 * 4, 0.05), (32768, 0.04), (1000, 0.0), (5000, 1.0), (1, 1.0), (65536, 0.5), (1025, 0.5), (70001, 0.3)):
        lab = torch.where(torch.rand(n, device=dev()) < frac, torch.randint(0, 30522, (n,), device=dev()), torch.full((n,), -1, device=dev()))
        idx = torch.full((n,), -7, device=dev(), dtype=torch.int32)
        cnt = torch.zeros(1, device=dev(), dtype=torch.int32)
        ops.masked_select(lab, idx, cnt)
        want = torch.nonzero(lab != -1).flatten().to(torch.int32)
        assert int(cnt.item()) == want.numel()
        assert torch.equal(idx[: want.numel()], want)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gather_scatter_rows(ops, dtype):
    from mvlt_amd._lib import rowmap
    B, HW, T, Cdim = 3, 10, 8, 512
    x = rnd(B, HW + T, Cdim, dtype=dtype)
    pos = torch.tensor([1, 5, 9, 17, 23], device=dev(), dtype=torch.int32)      # flat b*T+t
    out = torch.empty(5, Cdim, device=dev(), dtype=dtype)
    ops.gather_rows(x, pos, out, 5, Cdim, Cdim, src_map=rowmap(T, HW + T, HW))
    ref = x[:, HW:].reshape(B * T, Cdim)[pos.long()]
    assert torch.equal(out, ref)
    dst = torch.zeros_like(x)
    ops.scatter_rows(out, pos, dst, 5, Cdim, Cdim, dst_map=rowmap(T, HW + T, HW))
    assert torch.equal(dst[:, HW:].reshape(B * T, Cdim)[pos.long()], ref)
    assert dst.float().abs().sum().item() == ref.float().abs().sum().item()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("V,ld", [(30522, 30528), (2, 8), (122, 128)])
def test_cross_entropy(ops, dtype, V, ld):
    R = 37
    buf = rnd(R, ld, dtype=dtype, scale=3.0)
    labels = torch.randint(0, V, (R,), device=dev())
    labels[::5] = -1
    lse = torch.empty(R, device=dev())
    acc = torch.zeros(2, device=dev())
    ops.cross_entropy_fwd(buf, labels, lse, acc[0:1], acc[1:2], R, V, ld)
    lg = buf[:, :V].float().requires_grad_(True)
    ref = F.cross_entropy(lg, labels, ignore_index=-1)
    assert abs((acc[0] / acc[1]).item() - ref.item()) < 1e-4 * max(1, abs(ref.item()))
    ref.backward()
    dl = torch.full((R, ld), 7.0, device=dev(), dtype=torch.float32)
    gs = torch.tensor([1.0], device=dev())
    ops.cross_entropy_bwd(buf, labels, lse, gs, acc[1:2], dl, R, V, ld, ld)
    assert maxrel(dl[:, :V], lg.grad) < 1e-4
    assert dl[:, V:].abs().max().item() == 0.0


def test_adamw_matches_torch(ops):
    n = 4096 + 8
    p0 = rnd(n, dtype=torch.float32)
    p = p0.clone()
    g = rnd(n, dtype=torch.float32, seed=1)
    m, v = torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
    p16 = torch.empty(n, device=dev(), dtype=torch.bfloat16)
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref_p], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    for t in range(1, 4):
        ref_p.grad = g.clone()
        opt.step()
        hp = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 0.01, 1 - 0.9 ** t, 1 - 0.999 ** t, 1.0], device=dev())
        ops.adamw_step(p, g, m, v, p16, n, hp)
    assert maxrel(p, ref_p.detach()) < 1e-6
    assert torch.equal(p16, p.to(torch.bfloat16))
    # per-parameter weight-decay mask (timm's no-decay split for biases / 1-D tensors)
    mask = (torch.arange(n, device=dev()) % 3 == 0).to(torch.uint8)
    pa, pb = p0.clone(), p0.clone()
    ma, va, mb, vb = (torch.zeros(n, device=dev()) for _ in range(4))
    hp = torch.tensor([1e-3, 0.9, 0.999, 1e-8, 0.5, 0.1, 0.001, 1.0], device=dev())
    hp0 = hp.clone(); hp0[4] = 0.0
    ops.adamw_step(pa, g, ma, va, None, n, hp, mask)
    ops.adamw_step(pb, g, mb, vb, None, n, hp)
    pc = p0.clone(); mc, vc = torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
    ops.adamw_step(pc, g, mc, vc, None, n, hp0)
    assert torch.equal(pa[mask.bool()], pb[mask.bool()]) and torch.equal(pa[~mask.bool()], pc[~mask.bool()])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_transpose_cast(ops, dtype):
    R, Cc = 30522, 768
    w = rnd(R, Cc, dtype=torch.float32)
    ld = (R + 7) // 8 * 8
    out = torch.zeros(Cc, ld, device=dev(), dtype=dtype)
    ops.transpose_cast(w, out, R, Cc, ld_out=ld)
    assert torch.equal(out[:, :R], w.t().to(dtype))
    assert out[:, R:].abs().max().item() == 0


# ------------------------------------------------------------------ fused MLP (stages 1 / 2)
@pytest.mark.parametrize("Cdim,hid,M,Bsz", [(64, 512, 3 * 400, 3), (128, 1024, 2 * 333, 2), (64, 512, 130, 1),
                                            # samples of whole 64-token tiles (what the model's stages give: 4224 / 1152 rows per sample): the
                                            # round-3 weight-gradient kernel (tile-uniform DropPath factor, dropped samples skipped); the last
                                            # case has two DIFFERENT non-zero factors (the accumulator-rescale branch)
                                            (64, 512, 3 * 448, 3), (128, 1024, 3 * 320, 3), (64, 512, 4 * 192, 4)])
def test_fused_mlp(ops, Cdim, hid, M, Bsz):
    bf = torch.bfloat16
    x = rnd(M, Cdim, dtype=bf)
    w1, w2 = rnd(hid, Cdim, dtype=bf, seed=1, scale=Cdim ** -0.5), rnd(Cdim, hid, dtype=bf, seed=2, scale=hid ** -0.5)
    b1, b2 = 0.1 * rnd(hid, dtype=torch.float32, seed=3), 0.1 * rnd(Cdim, dtype=torch.float32, seed=4)
    res = rnd(M, Cdim, dtype=torch.float32, seed=5)
    scale = torch.tensor([1.0 / 0.9, 0.0, 1.0 / 0.9, 1.0 / 0.75][:Bsz], device=dev())
    rps = M // Bsz
    out = torch.empty(M, Cdim, device=dev())
    hbuf = torch.empty(M, hid, device=dev(), dtype=bf)
    ops.mlp_fwd(x, w1, b1, w2, b2, res, out, M, Cdim, hid, row_scale=scale, rows_per_scale=rps, h_out=hbuf)
    xr = x.float().requires_grad_(True)
    w1r, w2r = w1.float().requires_grad_(True), w2.float().requires_grad_(True)
    b1r, b2r = b1.clone().requires_grad_(True), b2.clone().requires_grad_(True)
    h = xr @ w1r.t() + b1r
    rowscale = scale.repeat_interleave(rps)[:, None]
    branch = (F.gelu(h) @ w2r.t() + b2r) * rowscale
    ref = branch + res
    assert maxrel(hbuf.float(), h) < 2e-2
    assert maxrel(out - res, branch) < 2e-2
    # without the pre-activation store the forward is the software-pipelined kernel (mlp_pipe_kernel): the launch the step makes
    out2 = torch.full((M, Cdim), float("nan"), device=dev())
    ops.mlp_fwd(x, w1, b1, w2, b2, res, out2, M, Cdim, hid, row_scale=scale, rows_per_scale=rps)
    assert maxrel(out2 - res, branch) < 2e-2
    dy = rnd(M, Cdim, dtype=bf, seed=7)
    ref.backward(dy.float())
    dx = torch.empty(M, Cdim, device=dev(), dtype=bf)
    ops.mlp_bwd_dx(x, dy, w1, w1.t().contiguous(), w2.t().contiguous(), b1, dx, M, Cdim, hid, row_scale=scale, rows_per_scale=rps)
    assert maxrel(dx.float(), xr.grad) < 3e-2
    dw1, db1 = torch.zeros(hid, Cdim, device=dev()), torch.zeros(hid, device=dev())
    dw2, db2 = torch.zeros(Cdim, hid, device=dev()), torch.zeros(Cdim, device=dev())
    ops.mlp_bwd_dw(x, dy, w1, w2.t().contiguous(), b1, dw1, db1, dw2, db2, M, Cdim, hid, row_scale=scale, rows_per_scale=rps)
    for a, r, nm in ((dw1, w1r.grad, "dw1"), (db1, b1r.grad, "db1"), (dw2, w2r.grad, "dw2"), (db2, b2r.grad, "db2")):
        assert maxrel(a, r) < 3e-2, nm
    # round 6: the same gradients with the token splits leaving as bf16 partial tiles + the ordered fold (mvlt_mlp_args.partials; taken by the tile-uniform kernel): accumulates
    # into dw1 / dw2 like the atomic path, equals it to the partial tiles' bf16 rounding, is bit-identical from launch to launch, folded at once or with the pending ones
    if rps % 64 == 0:
        from mvlt_amd._lib import last_kernel
        scratch = torch.empty(64 * 65536, device=dev(), dtype=bf)
        got = []
        for defer in (False, True, True):
            e1, eb1 = torch.full((hid, Cdim), 0.5, device=dev()), torch.zeros(hid, device=dev())
            e2, eb2 = torch.full((Cdim, hid), -0.25, device=dev()), torch.zeros(Cdim, device=dev())
            ops.mlp_bwd_dw(x, dy, w1, w2.t().contiguous(), b1, e1, eb1, e2, eb2, M, Cdim, hid, row_scale=scale, rows_per_scale=rps, partials=scratch, defer_fold=defer)
            if defer:
                torch.cuda.synchronize()
                assert bool((e1 == 0.5).all()) and bool((e2 == -0.25).all())            # nothing folded yet
                ops.tn_fold_flush(scratch)
            torch.cuda.synchronize()
            assert "tn_fold" in last_kernel()
            got.append((e1, e2))
            assert maxrel(e1 - 0.5, dw1) < 5e-3 and maxrel(e2 + 0.25, dw2) < 5e-3
            assert maxrel(eb1, db1) < 1e-3 and maxrel(eb2, db2) < 1e-3
        assert torch.equal(got[0][0], got[1][0]) and torch.equal(got[1][0], got[2][0]) and torch.equal(got[0][1], got[2][1])


@pytest.mark.parametrize("Cdim,hid,M,Bsz", [(64, 512, 3 * 400, 3), (128, 1024, 2 * 333, 2), (64, 512, 130, 1)])
def test_fused_mlp_dx_with_layernorm_backward(ops, Cdim, hid, M, Bsz):
    """mvlt_mlp_bwd_dx with lnb_x: norm2's backward from the dx kernel's epilogue (dx += LN backward in place, DropPath-scaled copy,
    parameter gradients through per-workgroup partials) against the two separate launches it replaces"""
    bf = torch.bfloat16
    xm = rnd(M, Cdim, dtype=torch.float32, scale=1.5) + 0.3
    g, b = 1 + 0.2 * rnd(Cdim, dtype=torch.float32, seed=1), 0.1 * rnd(Cdim, dtype=torch.float32, seed=2)
    xn = torch.empty(M, Cdim, device=dev(), dtype=bf)
    mean, rstd = torch.empty(M, device=dev()), torch.empty(M, device=dev())
    ops.layernorm_fwd(xm, xn, g, b, M, Cdim, Cdim, Cdim, 1e-6, mean=mean, rstd=rstd)
    w1, w2 = rnd(hid, Cdim, dtype=bf, seed=3, scale=Cdim ** -0.5), rnd(Cdim, hid, dtype=bf, seed=4, scale=hid ** -0.5)
    b1 = 0.1 * rnd(hid, dtype=torch.float32, seed=5)
    s2 = torch.tensor([1.0 / 0.9, 0.0, 1.0 / 0.9][:Bsz], device=dev())
    s1 = torch.tensor([0.0, 1.0 / 0.8, 1.0 / 0.8][:Bsz], device=dev())
    rps = M // Bsz
    dy = rnd(M, Cdim, dtype=bf, seed=7)                      # gradient w.r.t. the block output = the residual gradient stream
    w1t, w2t = w1.t().contiguous(), w2.t().contiguous()
    # reference: the two launches
    dxn = torch.empty(M, Cdim, device=dev(), dtype=bf)
    ops.mlp_bwd_dx(xn, dy, w1, w1t, w2t, b1, dxn, M, Cdim, hid, row_scale=s2, rows_per_scale=rps)
    dx_ref, dx2_ref = dy.clone(), torch.empty_like(dy)
    dg_ref, db_ref = torch.zeros(Cdim, device=dev()), torch.zeros(Cdim, device=dev())
    ops.layernorm_bwd(dxn, xm, dx_ref, g, mean, rstd, M, Cdim, Cdim, Cdim, Cdim, dgamma=dg_ref, dbeta=db_ref, accumulate=True,
                      dx2=dx2_ref, dx2_scale=s1, dx2_rows_per_scale=rps, lddx2=Cdim)
    # fused
    dx, dx2 = dy.clone(), torch.empty_like(dy)
    dg, db = torch.full((Cdim,), 2.0, device=dev()), torch.full((Cdim,), -1.0, device=dev())       # accumulate semantics
    ops.mlp_bwd_dx(xn, dx, w1, w1t, w2t, b1, None, M, Cdim, hid, row_scale=s2, rows_per_scale=rps,
                   ln_bwd=dict(x=xm, mean=mean, rstd=rstd, gamma=g, dx=dx, dgamma=dg, dbeta=db, dx2=dx2, dx2_scale=s1, dx2_rows_per_scale=rps))
    scale = (dx_ref.float() - dy.float()).abs().max().item()
    assert (dx.float() - dx_ref.float()).abs().max().item() <= 2e-2 * scale + 2 ** -7 * dy.float().abs().max().item()
    assert rel(dx.float() - dy.float(), dx_ref.float() - dy.float()) < 2e-2          # bf16 rounding of dxn in the two-launch path only
    assert rel(dx2.float(), dx2_ref.float()) < 2e-2
    assert maxrel(dg - 2.0, dg_ref) < 2e-2 and maxrel(db + 1.0, db_ref) < 2e-2
    # without the second output
    dx_b = dy.clone()
    dg_b, db_b = torch.zeros(Cdim, device=dev()), torch.zeros(Cdim, device=dev())
    ops.mlp_bwd_dx(xn, dx_b, w1, w1t, w2t, b1, None, M, Cdim, hid, row_scale=s2, rows_per_scale=rps,
                   ln_bwd=dict(x=xm, mean=mean, rstd=rstd, gamma=g, dx=dx_b, dgamma=dg_b, dbeta=db_b))
    assert torch.equal(dx_b, dx)


# ------------------------------------------------------------------ bilinear upsample (align_corners=True) and its adjoint
@pytest.mark.parametrize("B,H,C,s,nchw,out_dtype", [(2, 8, 64, 2, False, torch.float32), (3, 6, 192, 2, False, torch.bfloat16),
                                                    (2, 8, 3, 8, True, torch.float32), (1, 5, 6, 3, False, torch.float32),
                                                    (2, 32, 3, 8, True, torch.float32), (1, 5, 3, 3, True, torch.float32)])
def test_upsample_fwd_bwd(ops, B, H, C, s, nchw, out_dtype):
    """reference libs/vl_heads.py:128-134 (nn.Upsample(scale_factor, 'bilinear', align_corners=True)) on pixel-major tensors."""
    x = rnd(B * H * H, C, dtype=torch.float32)
    xt = x.view(B, H, H, C).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    ref = F.interpolate(xt, scale_factor=s, mode="bilinear", align_corners=True)
    Ho = H * s
    if nchw:
        out = torch.empty(B, C, Ho, Ho, device=dev(), dtype=torch.float32)
        ops.upsample_fwd(x, C, B, H, H, C, s, out, 0, nchw=True)
        got = out
    else:
        out = torch.empty(B * Ho * Ho, C, device=dev(), dtype=out_dtype)
        ops.upsample_fwd(x, C, B, H, H, C, s, out, C)
        got = out.float().view(B, Ho, Ho, C).permute(0, 3, 1, 2)
    assert maxrel(got, ref.detach()) < (1e-5 if out_dtype == torch.float32 else TOL[torch.bfloat16])
    g = rnd(B, C, Ho, Ho, dtype=torch.float32, seed=3)
    ref.backward(g)
    dy = g.contiguous() if nchw else g.permute(0, 2, 3, 1).reshape(B * Ho * Ho, C).contiguous()
    base = rnd(B * H * H, C, dtype=torch.float32, seed=5)
    dx = base.clone()
    ops.upsample_bwd(dy, 0 if nchw else C, nchw, B, H, H, C, s, dx, C, accumulate=True)
    want = base + xt.grad.permute(0, 2, 3, 1).reshape(B * H * H, C)
    assert maxrel(dx, want) < 1e-5
    if not nchw and C % 4 == 0:  # bf16 dy (a conv input gradient in the operand dtype): same adjoint on the rounded values
        dyh = dy.to(torch.bfloat16)
        dxa, dxb = torch.zeros(B * H * H, C, device=dev()), torch.zeros(B * H * H, C, device=dev())
        ops.upsample_bwd(dyh, C, False, B, H, H, C, s, dxa, C)
        ops.upsample_bwd(dyh.float(), C, False, B, H, H, C, s, dxb, C)
        assert torch.equal(dxa, dxb)
    if nchw:                     # bf16 dx, rows padded to 8 (the score conv's gradient operand, mim.py): pad columns stay untouched
        dx16 = torch.zeros(B * H * H, 8, device=dev(), dtype=torch.bfloat16)
        ops.upsample_bwd(dy, 0, True, B, H, H, C, s, dx16, 8)
        assert maxrel(dx16[:, :C].float(), want - base) < TOL[torch.bfloat16]
        assert (dx16[:, C:] == 0).all()


@pytest.mark.parametrize("B,H,C,s,out_dtype", [(2, 32, 3, 8, torch.bfloat16), (3, 8, 3, 8, torch.float32), (1, 12, 3, 8, torch.float32), (2, 16, 6, 4, torch.bfloat16)])
def test_upsample_smooth_l1_fused(ops, B, H, C, s, out_dtype):
    """mvlt_upsample_l1_fwd / _bwd: SmoothL1(mean) of the bilinear x s upsample (align_corners=True) against an NCHW target without the
    upsampled tensor (reference libs/vl_heads.py:163-165 + engine_grid_masking.py:99) -- against F.interpolate + F.smooth_l1_loss"""
    x = rnd(B * H * H, C, dtype=torch.float32, scale=1.5)
    Ho = H * s
    target = rnd(B, C, Ho, Ho, dtype=torch.float32, seed=3)
    xt = x.view(B, H, H, C).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    pred = F.interpolate(xt, scale_factor=s, mode="bilinear", align_corners=True)
    ref = F.smooth_l1_loss(pred, target)
    acc = torch.zeros(1, device=dev())
    ops.upsample_l1_fwd(x, C, B, H, H, C, s, target, acc)
    got = acc.item() / target.numel()
    assert abs(got - ref.item()) <= 1e-5 * abs(ref.item())
    (ref * 3.0).backward()
    want = xt.grad.permute(0, 2, 3, 1).reshape(B * H * H, C)
    ld = 8
    dx = torch.zeros(B * H * H, ld, device=dev(), dtype=out_dtype)
    ops.upsample_l1_bwd(x, C, B, H, H, C, s, target, torch.full((1,), 3.0, device=dev()), dx, ld)
    assert maxrel(dx[:, :C].float(), want) < (1e-5 if out_dtype == torch.float32 else TOL[torch.bfloat16])
    assert (dx[:, C:] == 0).all()


# ------------------------------------------------------------------ train-mode BatchNorm over pixel-major matrices
@pytest.mark.parametrize("M,C,lddy,off", [(5000, 64, 192, 0), (3001, 192, 192, 0), (2048, 128, 192, 64), (777, 6, 6, 0)])
def test_batchnorm_fwd_bwd(ops, M, C, lddy, off):
    """reference libs/vl_heads.py:113-125 (BasicConv2d's nn.BatchNorm2d in train mode) on [pixels, channels] fp32; dy may be a
    column range of a wider matrix (the concat gradients)."""
    z = rnd(M, C, dtype=torch.float32, scale=2.0) + 0.5
    gamma, beta = rnd(C, dtype=torch.float32, seed=1) + 1.0, rnd(C, dtype=torch.float32, seed=2)
    dyw = rnd(M, lddy, dtype=torch.float32, seed=3)
    dy = dyw[:, off:off + C]
    # fp64 autograd of the defining formulas (MIOpen's own batch-norm backward is not accurate enough to be the reference here)
    zr = z.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = (zr - zr.mean(0)) * (zr.var(0, unbiased=False) + 1e-5).rsqrt() * gr + br
    ref.backward(dy.double())
    s = torch.zeros(2, C, device=dev())
    ops.col_stats(z, C, M, C, s[0], s[1])
    mean, rstd = torch.empty(C, device=dev()), torch.empty(C, device=dev())
    ops.bn_finalize(s[0], s[1], M, C, 1e-5, 0.1, mean, rstd)
    assert maxrel(mean, z.mean(0)) < 1e-5 and maxrel(rstd, (z.var(0, unbiased=False) + 1e-5).rsqrt()) < 1e-4
    if C % 4 == 0:
        y = torch.empty(M, C, device=dev())
        ops.bn_norm(z, C, mean, rstd, gamma, beta, M, C, y32=y, ld32=C)
        assert maxrel(y, ref.detach()) < 1e-4
    red = torch.zeros(2, C, device=dev())
    ops.bn_bwd_reduce(dy, lddy, z, C, mean, rstd, M, C, red[0], red[1])
    assert maxrel(red[0], br.grad) < 1e-4 and maxrel(red[1], gr.grad) < 1e-4
    if C % 4 == 0:
        gb, gg = torch.ones(C, device=dev()), torch.full((C,), 2.0, device=dev())
        dz = torch.empty(M, C, device=dev())
        ops.bn_bwd_apply(dy, lddy, z, C, mean, rstd, gamma, red[0], red[1], M, C, dz, C, g_beta=gb, g_gamma=gg)
        assert maxrel(dz, zr.grad) < 1e-3
        assert maxrel(gb - 1.0, br.grad) < 1e-4 and maxrel(gg - 2.0, gr.grad) < 1e-4
        # bf16 dy (what the MIM decoder's first backward stages hand over): same kernels reading half the bytes; checked against the fp32
        # kernels fed the bf16-rounded gradient
        dyh = dyw.to(torch.bfloat16)[:, off:off + C]
        dyr = dyh.float()
        r32, r16 = torch.zeros(2, C, device=dev()), torch.zeros(2, C, device=dev())
        ops.bn_bwd_reduce(dyr.contiguous(), C, z, C, mean, rstd, M, C, r32[0], r32[1])
        ops.bn_bwd_reduce(dyh, lddy, z, C, mean, rstd, M, C, r16[0], r16[1])
        assert maxrel(r16, r32) < 1e-5
        d32, d16 = torch.empty(M, C, device=dev()), torch.empty(M, C, device=dev())
        ops.bn_bwd_apply(dyr.contiguous(), C, z, C, mean, rstd, gamma, r32[0], r32[1], M, C, d32, C)
        ops.bn_bwd_apply(dyh, lddy, z, C, mean, rstd, gamma, r32[0], r32[1], M, C, d16, C)
        assert torch.equal(d16, d32)
        # fp16 z (the bf16 path's conv output): the same kernels reading half the bytes == the fp32 kernels fed the fp16-rounded z
        zh = z.to(torch.float16)
        zr32 = zh.float()
        yb, yh = torch.empty(M, C, device=dev(), dtype=torch.bfloat16), torch.empty(M, C, device=dev(), dtype=torch.bfloat16)
        y32b, y32h = torch.empty(M, C, device=dev()), torch.empty(M, C, device=dev())
        ops.bn_norm(zr32, C, mean, rstd, gamma, beta, M, C, y32=y32b, ld32=C, y16=yb, ld16=C)
        ops.bn_norm(zh, C, mean, rstd, gamma, beta, M, C, y32=y32h, ld32=C, y16=yh, ld16=C)
        assert torch.equal(yb, yh) and torch.equal(y32b, y32h)
        if C % 8 == 0:
            yf16 = torch.empty(M, C, device=dev(), dtype=torch.float16)            # fp16 "wide" output (the factors of the decoder's feature product)
            ops.bn_norm(zh, C, mean, rstd, gamma, beta, M, C, y32=yf16, ld32=C)
            assert torch.equal(yf16, y32h.to(torch.float16))
        for dyv, ldv in ((dyh, lddy), (dy, lddy)):
            ra, rb = torch.zeros(2, C, device=dev()), torch.zeros(2, C, device=dev())
            ops.bn_bwd_reduce(dyv, ldv, zr32, C, mean, rstd, M, C, ra[0], ra[1])
            ops.bn_bwd_reduce(dyv, ldv, zh, C, mean, rstd, M, C, rb[0], rb[1])
            assert maxrel(rb, ra) < 1e-5
            da, db = torch.empty(M, C, device=dev(), dtype=torch.bfloat16), torch.empty(M, C, device=dev(), dtype=torch.bfloat16)
            ops.bn_bwd_apply(dyv, ldv, zr32, C, mean, rstd, gamma, ra[0], ra[1], M, C, da, C)
            ops.bn_bwd_apply(dyv, ldv, zh, C, mean, rstd, gamma, ra[0], ra[1], M, C, db, C)
            assert torch.equal(da, db)


@pytest.mark.parametrize("M,C,copies", [(4096, 64, 16), (3000, 192, 4), (777, 128, 1)])
def test_batchnorm_finalize_inside_the_normalise_launch(ops, M, C, copies):
    """mvlt_bn_finalize_norm == mvlt_bn_finalize followed by mvlt_bn_norm, bit for bit: statistics, running statistics, both outputs (fp16 z; reference
    libs/vl_heads.py:113-125)."""
    z = (rnd(M, C, dtype=torch.float32, scale=2.0) + 0.5).to(torch.float16)
    gamma, beta = rnd(C, dtype=torch.float32, seed=1) + 1.0, rnd(C, dtype=torch.float32, seed=2)
    zf = z.float()
    parts = torch.zeros(2, copies, C, device=dev())
    for k in range(copies):                                   # the conv epilogue's interleaved accumulators: row tile t adds into copy t % copies
        rows = zf[k::copies]
        parts[0, k], parts[1, k] = rows.sum(0), (rows * rows).sum(0)
    outs = []
    for fused in (False, True):
        mean, rstd = torch.empty(C, device=dev()), torch.empty(C, device=dev())
        rm, rv = torch.full((C,), 0.25, device=dev()), torch.full((C,), 1.5, device=dev())
        y32, y16 = torch.empty(M, C, device=dev(), dtype=torch.float16), torch.empty(M, 2 * C, device=dev(), dtype=torch.bfloat16)
        if fused:
            ops.bn_finalize_norm(z, C, parts[0], parts[1], copies, 1e-5, 0.1, mean, rstd, rm, rv, gamma, beta, M, C, y32=y32, ld32=C, y16=y16[:, C:], ld16=2 * C)
        else:
            ops.bn_finalize(parts[0], parts[1], M, C, 1e-5, 0.1, mean, rstd, rm, rv, copies=copies)
            ops.bn_norm(z, C, mean, rstd, gamma, beta, M, C, y32=y32, ld32=C, y16=y16[:, C:], ld16=2 * C)
        outs.append((mean, rstd, rm, rv, y32, y16[:, C:].clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert maxrel(outs[1][0], zf.mean(0)) < 1e-4 and maxrel(outs[1][1], (zf.var(0, unbiased=False) + 1e-5).rsqrt()) < 1e-3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_row_scale(ops, dtype):
    B, N, C = 5, 37, 64
    x = rnd(B * N, C, dtype=dtype)
    sc = torch.tensor([0.0, 1.25, 1.0, 0.0, 1.111], device=dev())
    out = torch.empty_like(x)
    ops.row_scale(x, sc, N, B * N, C, out)
    want = (x.float().view(B, N, C) * sc.view(B, 1, 1)).view(B * N, C).to(dtype)
    assert torch.equal(out, want)


def test_smooth_l1_matches_torch(ops):
    """reference engine_grid_masking.py:99: F.smooth_l1_loss(t2i_logits, images), forward value and gradient."""
    from mvlt_amd.engine import smooth_l1
    pred = (rnd(3, 3, 64, 64, dtype=torch.float32) * 1.5).requires_grad_(True)
    target = rnd(3, 3, 64, 64, dtype=torch.float32, seed=4)
    loss = smooth_l1(pred, target)
    (loss * 10).backward()
    pr = pred.detach().clone().requires_grad_(True)
    ref = F.smooth_l1_loss(pr, target)
    (ref * 10).backward()
    assert abs(loss.item() - ref.item()) < 1e-6 * max(1.0, abs(ref.item()))
    assert maxrel(pred.grad, pr.grad) < 1e-6


def test_ew_mul3_bwd(ops):
    """gradients of the three-way feature product (reference libs/vl_heads.py:152), dy read as a column range of a wider matrix"""
    M, Cd = 3000, 64
    a, b, c = (rnd(M, Cd, dtype=torch.float32, seed=s) for s in (1, 2, 3))
    dyw = rnd(M, 3 * Cd, dtype=torch.float32, seed=4)
    da, db, dc = (torch.empty(M, Cd, device=dev()) for _ in range(3))
    ops.ew_mul3_bwd(dyw, 3 * Cd, a, b, c, Cd, da, db, dc, M, Cd)
    dy = dyw[:, :Cd]
    assert maxrel(da, dy * b * c) < 1e-6 and maxrel(db, dy * a * c) < 1e-6 and maxrel(dc, dy * a * b) < 1e-6
    dyh = dyw.to(torch.bfloat16)
    dah, dbh, dch = (torch.empty(M, Cd, device=dev(), dtype=torch.bfloat16) for _ in range(3))
    ops.ew_mul3_bwd(dyh, 3 * Cd, a, b, c, Cd, dah, dbh, dch, M, Cd)
    dy = dyh[:, :Cd].float()
    assert torch.equal(dah, (dy * b * c).to(torch.bfloat16)) and torch.equal(dbh, (dy * a * c).to(torch.bfloat16)) and torch.equal(dch, (dy * a * b).to(torch.bfloat16))
    # fp16 factors (what the bf16 training path keeps of low / cu2o / cu3o): the same kernels fed the fp16-rounded factors; forward product too
    ah, bh, ch = a.to(torch.float16), b.to(torch.float16), c.to(torch.float16)
    d2 = [torch.empty(M, Cd, device=dev(), dtype=torch.bfloat16) for _ in range(3)]
    d3 = [torch.empty(M, Cd, device=dev(), dtype=torch.bfloat16) for _ in range(3)]
    ops.ew_mul3_bwd(dyh, 3 * Cd, ah, bh, ch, Cd, *d2, M, Cd)
    ops.ew_mul3_bwd(dyh, 3 * Cd, ah.float(), bh.float(), ch.float(), Cd, *d3, M, Cd)
    assert all(torch.equal(x, y) for x, y in zip(d2, d3))
    p16, p32 = torch.empty(M, 3 * Cd, device=dev(), dtype=torch.bfloat16), torch.empty(M, 3 * Cd, device=dev(), dtype=torch.bfloat16)
    ops.ew_mul(None, 0, ah, Cd, bh, Cd, ch, Cd, M=M, Cdim=Cd, out16=p16, ld16=3 * Cd)
    ops.ew_mul(None, 0, ah.float(), Cd, bh.float(), Cd, ch.float(), Cd, M=M, Cdim=Cd, out16=p32, ld16=3 * Cd)
    assert torch.equal(p16[:, :Cd], p32[:, :Cd])


# ------------------------------------------------------------------ round 2 additions
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("r,Hin,Cin,Cout", [(2, 8, 64, 128), (8, 16, 64, 64), (4, 8, 128, 128)])
def test_gemm_tn_conv_weight_layout(ops, dtype, r, Hin, Cin, Cout):
    """c_taps / c_seg: the weight gradient of a kernel==stride conv lands in nn.Conv2d's [out][cin][kh][kw] layout directly"""
    from mvlt_amd._lib import patchmap
    Bsz, T = 2, 5
    HWi, Ho = Hin * Hin, Hin // r
    X = rnd(Bsz, HWi + T, Cin, dtype=dtype)
    M, K = Bsz * Ho * Ho, r * r * Cin
    dY = rnd(M, Cout, dtype=dtype, seed=3)
    dW = torch.zeros(Cout, Cin, r, r, device=dev(), dtype=torch.float32)
    cs = torch.zeros(Cout, device=dev(), dtype=torch.float32)
    ops.gemm_tn(dY, X, dW.view(Cout, K), M, Cout, K, Cout, Cin, K, b_map=patchmap(r, Hin, HWi + T, Ho * Ho, Ho, Cin), colsum=cs, taps=r * r, seg=Cin)
    img = X[:, :HWi].float().transpose(1, 2).reshape(Bsz, Cin, Hin, Hin)
    Wc = torch.zeros(Cout, Cin, r, r, device=dev(), requires_grad=True)
    bc = torch.zeros(Cout, device=dev(), requires_grad=True)
    y = F.conv2d(img, Wc, bc, stride=r).flatten(2).transpose(1, 2).reshape(M, Cout)
    (y * dY.float()).sum().backward()
    assert maxrel(dW, Wc.grad) < TOL[dtype]
    assert maxrel(cs, bc.grad) < TOL[dtype]


def test_gemm_tn_conv3x3_weight_layout(ops):
    from mvlt_amd._lib import conv3map
    Bsz, side, Cin, Cout = 2, 8, 64, 128
    M = Bsz * side * side
    X = rnd(M, Cin, dtype=torch.bfloat16)
    dY = rnd(M, Cout, dtype=torch.bfloat16, seed=2)
    dW = torch.zeros(Cout, Cin, 3, 3, device=dev(), dtype=torch.float32)
    ops.gemm_tn(dY, X, dW.view(Cout, 9 * Cin), M, Cout, 9 * Cin, Cout, Cin, 9 * Cin, b_map=conv3map(side, side, side * side, Cin), taps=9, seg=Cin)
    img = X.float().reshape(Bsz, side, side, Cin).permute(0, 3, 1, 2)
    Wc = torch.zeros(Cout, Cin, 3, 3, device=dev(), requires_grad=True)
    y = F.conv2d(img, Wc, None, padding=1).permute(0, 2, 3, 1).reshape(M, Cout)
    (y * dY.float()).sum().backward()
    assert maxrel(dW, Wc.grad) < TOL[torch.bfloat16]


@pytest.mark.parametrize("gin,gout,C,skip", [(56, 64, 64, 0), (28, 48, 128, 0), (7, 12, 512, 1), (14, 16, 320, 0), (3, 2, 64, 0)])
def test_pos_embed_resize_matches_interpolate(ops, gin, gout, C, skip):
    """mvlt_resize_bilinear_tokens vs F.interpolate(mode='bilinear') (align_corners=False) on a (1, n, C) position embedding
    (reference libs/pvlt.py:291-297), forward and adjoint; skip=1: stage 4 skips the leading cls slot"""
    param = rnd(1, gin * gin + skip, C, dtype=torch.float32)
    pe = param[:, skip:][0]
    out = torch.empty(gout * gout, C, device=dev())
    ops.resize_bilinear_tokens(pe, out, gin, gin, gout, gout, C)
    pr = param.clone().requires_grad_(True)
    t = pr[:, skip:].reshape(1, gin, gin, C).permute(0, 3, 1, 2)
    ref = F.interpolate(t, size=(gout, gout), mode="bilinear").reshape(1, C, gout * gout).permute(0, 2, 1)[0]
    assert (out - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    dy = rnd(gout * gout, C, dtype=torch.float32, seed=7)
    ref.backward(dy)
    g = torch.zeros_like(param)
    g[:, :skip] = 3.0                                             # untouched slot
    ops.resize_bilinear_tokens(dy, g[:, skip:][0], gin, gin, gout, gout, C, adjoint=True)
    assert maxrel(g[:, skip:], pr.grad[:, skip:]) < 1e-5
    assert skip == 0 or float(g[0, 0, 0]) == 3.0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gelu_bwd(ops, dtype):
    dy, h = rnd(777, 768, dtype=dtype), rnd(777, 768, dtype=dtype, seed=1, scale=2.0)
    out = ops.gelu_bwd(dy, h, torch.empty_like(dy))
    hr = h.float().requires_grad_(True)
    F.gelu(hr).backward(dy.float())
    assert maxrel(out.float(), hr.grad) < TOL[dtype]


def test_small_head_cross_entropy_fn():
    """engine.cross_entropy (HIP row kernels behind autograd) vs F.cross_entropy for the ITM / CLS heads"""
    from mvlt_amd.engine import cross_entropy
    for V in (2, 48, 122):
        lg = rnd(37, V, dtype=torch.float32, scale=3.0).requires_grad_(True)
        lab = torch.randint(0, V, (37,), device=dev())
        loss = cross_entropy(lg, lab)
        (loss * 1.7).backward()
        lr = lg.detach().clone().requires_grad_(True)
        ref = F.cross_entropy(lr, lab)
        (ref * 1.7).backward()
        assert abs(float(loss) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
        assert maxrel(lg.grad, lr.grad) < 1e-5


@pytest.mark.parametrize("Cdim,hid,M,Bsz", [(64, 512, 3 * 400, 3), (128, 1024, 2 * 333, 2), (64, 512, 130, 1)])
def test_fused_mlp_with_folded_layernorm(ops, Cdim, hid, M, Bsz):
    """mvlt_mlp_fwd with ln_x: LayerNorm(norm2, eps 1e-6) folded into the operand load must equal mvlt_layernorm_fwd followed by the
    plain fused MLP -- same output, same stored LN output and row statistics"""
    bf = torch.bfloat16
    xm = rnd(M, Cdim, dtype=torch.float32, scale=1.5) + 0.3
    g, b = 1 + 0.2 * rnd(Cdim, dtype=torch.float32, seed=1), 0.1 * rnd(Cdim, dtype=torch.float32, seed=2)
    w1 = (rnd(hid, Cdim, dtype=torch.float32, seed=3) * Cdim ** -0.5).to(bf)
    w2 = (rnd(Cdim, hid, dtype=torch.float32, seed=4) * hid ** -0.5).to(bf)
    b1, b2 = 0.1 * rnd(hid, dtype=torch.float32, seed=5), 0.1 * rnd(Cdim, dtype=torch.float32, seed=6)
    rs = (torch.arange(Bsz, device=dev()) % 2).float() * 1.25
    xn = torch.empty(M, Cdim, device=dev(), dtype=bf)
    mean, rstd = torch.empty(M, device=dev()), torch.empty(M, device=dev())
    ops.layernorm_fwd(xm, xn, g, b, M, Cdim, Cdim, Cdim, 1e-6, mean=mean, rstd=rstd)
    ref = ops.mlp_fwd(xn, w1, b1, w2, b2, xm, torch.empty_like(xm), M, Cdim, hid, row_scale=rs, rows_per_scale=M // Bsz)
    xn2 = torch.empty_like(xn)
    mean2, rstd2 = torch.empty_like(mean), torch.empty_like(rstd)
    out = ops.mlp_fwd(None, w1, b1, w2, b2, xm, torch.empty_like(xm), M, Cdim, hid, row_scale=rs, rows_per_scale=M // Bsz,
                      ln=(g, b, 1e-6, xn2, mean2, rstd2))
    assert maxrel(mean2, mean) < 1e-5 and maxrel(rstd2, rstd) < 1e-5
    assert (xn2.float() - xn.float()).abs().max().item() <= 2 ** -7 * xn.float().abs().max().item()      # at most one bf16 ulp apart
    assert maxrel(out, ref) < 5e-3
    tref = F.layer_norm(xm, (Cdim,), g, b, 1e-6)
    assert maxrel(xn2.float(), tref) < TOL[bf]
    # out_op: the bf16 copy of the output next to (or instead of) the fp32 stream -- exactly the rounded fp32 values
    o16a, o16b = torch.empty_like(xn), torch.empty_like(xn)
    out_b = ops.mlp_fwd(None, w1, b1, w2, b2, xm, torch.empty_like(xm), M, Cdim, hid, row_scale=rs, rows_per_scale=M // Bsz,
                        ln=(g, b, 1e-6, xn2, mean2, rstd2), out_op=o16a)
    got = ops.mlp_fwd(None, w1, b1, w2, b2, xm, None, M, Cdim, hid, row_scale=rs, rows_per_scale=M // Bsz,
                      ln=(g, b, 1e-6, xn2, mean2, rstd2), out_op=o16b)
    assert got is o16b and torch.equal(out_b, out)
    assert torch.equal(o16a, out.to(bf)) and torch.equal(o16b, o16a)
    # post_ln: the NEXT block's norm1 of the output rows from the epilogue == mvlt_layernorm_fwd of the stored output
    g3, b3 = 1 + 0.2 * rnd(Cdim, dtype=torch.float32, seed=7), 0.1 * rnd(Cdim, dtype=torch.float32, seed=8)
    yn, mn, rn = torch.empty_like(xn), torch.empty_like(mean), torch.empty_like(rstd)
    out_c = ops.mlp_fwd(None, w1, b1, w2, b2, xm, torch.empty_like(xm), M, Cdim, hid, row_scale=rs, rows_per_scale=M // Bsz,
                        ln=(g, b, 1e-6, xn2, mean2, rstd2), post_ln=(g3, b3, 1e-6, yn, mn, rn))
    assert torch.equal(out_c, out)
    yr, mr, rr = torch.empty_like(xn), torch.empty_like(mean), torch.empty_like(rstd)
    ops.layernorm_fwd(out, yr, g3, b3, M, Cdim, Cdim, Cdim, 1e-6, mean=mr, rstd=rr)
    assert maxrel(mn, mr) < 1e-5 and maxrel(rn, rr) < 1e-5
    assert (yn.float() - yr.float()).abs().max().item() <= 2 ** -7 * yr.float().abs().max().item()


@pytest.mark.parametrize("side,Cin,Cout,Bsz,tokens_extra", [(32, 64, 64, 3, 0), (32, 192, 192, 2, 0), (16, 128, 64, 5, 128), (8, 64, 128, 9, 128),
                                                           (32, 128, 64, 1, 128), (16, 64, 64, 4, 0), (64, 64, 64, 2, 0), (8, 64, 64, 1, 0),
                                                           (24, 64, 64, 2, 0)])      # 24: not a halo width -> generic gathered GEMM
def test_conv3x3_wgrad_lds_halo(ops, side, Cin, Cout, Bsz, tokens_extra):
    """conv3_wgrad_kernel (mvlt_gemm_tn with the 3x3 gather on B, W in {8, 16, 32}): the nine taps' fragments come from ONE LDS-resident
    halo per 64-pixel k-tile; checked against autograd of F.conv2d, accumulate semantics, and the generic gathered TN GEMM."""
    import os
    from mvlt_amd._lib import conv3map
    tokens_in = side * side + tokens_extra                     # stage buffers carry the text tokens behind the image tokens
    X = rnd(Bsz, tokens_in, Cin, dtype=torch.bfloat16)
    M = Bsz * side * side
    dY = rnd(M, Cout, dtype=torch.bfloat16, seed=2)
    dW = torch.zeros(Cout, 9 * Cin, device=dev(), dtype=torch.float32)
    amap = conv3map(side, side, tokens_in, Cin)
    ops.gemm_tn(dY, X, dW, M, Cout, 9 * Cin, Cout, Cin, 9 * Cin, b_map=amap)
    img = X[:, : side * side].float().reshape(Bsz, side, side, Cin).permute(0, 3, 1, 2)
    Wc = torch.zeros(Cout, Cin, 3, 3, device=dev(), requires_grad=True)
    y = F.conv2d(img, Wc, None, padding=1).permute(0, 2, 3, 1).reshape(M, Cout)
    (y * dY.float()).sum().backward()
    ref = Wc.grad.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin)      # [out][tap][cin]
    assert maxrel(dW, ref) < TOL[torch.bfloat16]
    assert rel(dW, ref) < 5e-3
    ops.gemm_tn(dY, X, dW, M, Cout, 9 * Cin, Cout, Cin, 9 * Cin, b_map=amap)
    assert rel(dW, 2 * ref) < 5e-3
    # the same through bf16 partial tiles + the ordered fold instead of atomics (mvlt_gemm_tn_args.partials): accumulates, agrees, and repeats bit for bit
    from mvlt_amd._lib import last_kernel
    scratch = torch.empty(512 * 65536, device=dev(), dtype=torch.bfloat16)
    outs = []
    for _ in range(2):
        dWp = torch.full_like(dW, 0.5)
        ops.gemm_tn(dY, X, dWp, M, Cout, 9 * Cin, Cout, Cin, 9 * Cin, b_map=amap, partials=scratch)
        torch.cuda.synchronize()
        outs.append(dWp)
    if "tn_fold_kernel" in last_kernel():                     # (few pixels = fewer than four splits: the atomic flush stays)
        assert torch.equal(outs[0], outs[1])
    assert rel(outs[0] - 0.5, ref) < 6e-3


@pytest.mark.parametrize("side,Cin,Cout,Bsz,tokens_extra,variant",
                         [(32, 192, 192, 2, 0, "stats"), (32, 192, 192, 1, 0, "plain"), (32, 128, 128, 2, 128, "plain"), (32, 64, 64, 3, 0, "stats"),
                          (16, 128, 64, 5, 128, "plain"), (16, 64, 320, 4, 0, "strided"), (64, 64, 64, 1, 0, "acc"), (32, 64, 128, 2, 0, "acc"),
                          (16, 320, 64, 2, 0, "plain"),          # 320 channels: 5 slices
                          (16, 128, 128, 3, 0, "plain"), (16, 192, 192, 2, 128, "stats"), (32, 128, 256, 1, 0, "plain"), (32, 64, 192, 2, 0, "strided"),      # 8-wave tiles: W = 16, two N tiles, batch-strided rows
                          (32, 64, 128, 3, 0, "stats"), (16, 64, 128, 5, 0, "acc"),
                          (8, 64, 64, 3, 0, "plain"), (24, 64, 64, 2, 0, "plain"), (32, 72, 64, 1, 0, "plain")])      # not halo shapes -> generic gather
def test_conv3x3_nt_lds_halo(ops, side, Cin, Cout, Bsz, tokens_extra, variant):
    """conv3_nt_kernel (mvlt_gemm_nt with the 3x3 gather on A, W in {16, 32, 64}, 64-multiples of channels): the nine taps read their A
    fragments from one LDS-resident halo per 64-channel slice.  Checked against F.conv2d for the epilogues the MIM decoder uses: plain
    store, BatchNorm column statistics, accumulate into R, and batch-strided output rows."""
    from mvlt_amd._lib import conv3map, rowmap
    bf = torch.bfloat16
    tokens_in = side * side + tokens_extra
    X = rnd(Bsz, tokens_in, Cin, dtype=bf)
    Wk = rnd(Cout, 9 * Cin, dtype=bf, scale=0.05, seed=3)                     # [out][tap][cin]
    M = Bsz * side * side
    amap = conv3map(side, side, tokens_in, Cin)
    img = X[:, : side * side].float().reshape(Bsz, side, side, Cin).permute(0, 3, 1, 2)
    Wc = Wk.float().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    ref = F.conv2d(img, Wc, None, padding=1).permute(0, 2, 3, 1).reshape(M, Cout)
    if variant == "plain":
        out = torch.empty(M, Cout, device=dev(), dtype=torch.float32)
        ops.gemm_nt(X, Wk, out, M, Cout, 9 * Cin, Cin, 9 * Cin, Cout, a_map=amap)
        assert maxrel(out, ref) < TOL[bf]
        out16 = torch.empty(M, Cout, device=dev(), dtype=bf)
        ops.gemm_nt(X, Wk, out16, M, Cout, 9 * Cin, Cin, 9 * Cin, Cout, a_map=amap)
        assert maxrel(out16.float(), ref) < 2 * TOL[bf]
    elif variant == "stats":
        out = torch.empty(M, Cout, device=dev(), dtype=torch.float32)
        st = torch.zeros(2, 4, Cout, device=dev())
        ops.gemm_nt(X, Wk, out, M, Cout, 9 * Cin, Cin, 9 * Cin, Cout, a_map=amap, col_sum=st[0], col_sumsq=st[1], col_copies=4)
        assert maxrel(out, ref) < TOL[bf]
        assert maxrel(st[0].sum(0), out.sum(0)) < 1e-4
        assert maxrel(st[1].sum(0), (out * out).sum(0)) < 1e-4
        o16 = torch.empty(M, Cout, device=dev(), dtype=torch.float16)          # the same launch writing fp16 (what the decoder's bf16 path stores)
        s16 = torch.zeros(2, 4, Cout, device=dev())
        ops.gemm_nt(X, Wk, o16, M, Cout, 9 * Cin, Cin, 9 * Cin, Cout, a_map=amap, col_sum=s16[0], col_sumsq=s16[1], col_copies=4)
        assert torch.equal(o16, out.to(torch.float16))
        assert maxrel(s16[0].sum(0), o16.float().sum(0)) < 1e-4 and maxrel(s16[1].sum(0), (o16.float() ** 2).sum(0)) < 1e-4
    elif variant == "acc":
        base = rnd(M, Cout, dtype=torch.float32, seed=5)
        out = base.clone()
        ops.gemm_nt(X, Wk, out, M, Cout, 9 * Cin, Cin, 9 * Cin, Cout, a_map=amap, R=out)
        assert maxrel(out, ref + base) < TOL[bf]
    else:                                                                 # rows of image b land at b * stride + offset of a wider buffer
        stride, off = side * side + 7, 3
        buf = torch.full((Bsz * stride, Cout), 7.0, device=dev(), dtype=bf)
        ops.gemm_nt(X, Wk, buf, M, Cout, 9 * Cin, Cin, 9 * Cin, Cout, a_map=amap, c_map=rowmap(side * side, stride, off))
        got = buf.view(Bsz, stride, Cout)[:, off: off + side * side].reshape(M, Cout).float()
        assert maxrel(got, ref) < 2 * TOL[bf]
        assert (buf.view(Bsz, stride, Cout)[:, :off] == 7.0).all() and (buf.view(Bsz, stride, Cout)[:, off + side * side:] == 7.0).all()
