"""MIM decoder ("ITGHead", reference libs/vl_heads.py:107-165) as an explicit HIP kernel schedule.

Everything is pixel-major [M = B*H*W, C]: each conv3x3(pad 1) is mvlt_gemm_nt with the 3x3 neighbourhood row map
(im2col never exists in memory), its dgrad is the same gather with flipped/transposed taps, its wgrad is mvlt_gemm_tn with
the gather on the B operand.  BatchNorm uses the batch statistics of the local batch (train mode; the reference does
not convert to SyncBN) or the running statistics (eval), in fp32; the align_corners=True resizes and the three-way
feature products are fp32 kernels of csrc/mim.hip.  torch.cat is replaced by writing into column slices of the
concatenated buffers.  MFMA operands are the only tensors in the compute dtype.
"""
import os

import torch

from . import ops
from .params import pool_zeros
from ._lib import conv3map, rowmap

BN_EPS, BN_MOM = 1e-5, 0.1
_SEPARATE_STATS = bool(__import__('os').environ.get('MVLT_MIM_SEPARATE_STATS'))   # A/B switch: statistics by a second pass over z
STAT_COPIES = int(os.environ.get("MVLT_MIM_STAT_COPIES", "16"))           # interleaved batch-statistic accumulators of the conv epilogue (see mvlt_gemm_nt_args.col_copies)
CONVS = ("reduction1", "reduction2", "reduction3", "conv_upsample1", "conv_upsample2", "conv_upsample3", "conv_upsample4",
         "conv_upsample5", "conv_concat2", "conv_concat3", "conv4")


_FP32_DY = bool(os.environ.get("MVLT_MIM_FP32_DY"))      # A/B switch: every gradient map of the decoder's backward in fp32
_FP32_Z = bool(os.environ.get("MVLT_MIM_FP32_Z"))        # A/B switch: the pre-BatchNorm conv outputs stay fp32 on the bf16 path (rounds 1-3)
_NO_FIN_FUSE = bool(os.environ.get("MVLT_MIM_NO_FIN_FUSE"))   # A/B switch: BatchNorm statistics finalised by their own launch
_NO_BN_FOLD = bool(os.environ.get("MVLT_MIM_NO_BN_FOLD"))   # A/B switch: eval mode keeps the separate BatchNorm pass over an fp32 z


def _z(shape, dev, dtype=torch.float32):
    return pool_zeros(shape, dtype, dev)                          # step-scoped scratch (params.ZeroPool)


def _e(shape, dev, dtype=torch.float32):
    return torch.empty(shape, device=dev, dtype=dtype)


class MimStep:
    def __init__(self, model, x2, x3, x4, sides, training, need_grad):
        self.m, self.S = model, model.store
        self.dt = model.compute_dtype
        self.dev = x2.device
        self.x = (x2, x3, x4)                       # (B, N_i, C_i) stage outputs in the compute dtype, image tokens first
        self.B = x2.shape[0]
        self.s1, self.s2, self.s3 = sides           # 32, 16, 8 at 256 px
        self.M1, self.M2, self.M3 = (self.B * s * s for s in sides)
        self.training = training
        self.need_grad = need_grad
        self.rec = {}                               # per conv: saved tensors for backward
        self.nbt = []                               # num_batches_tracked buffers of the BatchNorms run in train mode

    # ---- one conv3x3 (no bias) + BatchNorm: returns the record; y is produced by `norm`
    def conv_bn(self, name, xin, ld_in, tokens_in, side, cin, cout, M):
        S, dev = self.S, self.dev
        p = f"t2i_head.{name}"
        amap = conv3map(side, side, tokens_in, cin)
        bn = getattr(self.m.t2i_head, name)[1]
        if not self.training and not self.need_grad and not _NO_BN_FOLD:
            # inference: BatchNorm on its running statistics is an affine map per output channel -- folded into the conv (`norm` launches it
            # with the destination it is given): no fp32 z, no normalisation pass, eleven launches fewer per forward
            r = dict(name=name, p=p, fold=self._folded(name, p, bn, cin, cout), xin=xin, ld_in=ld_in, amap=amap, cin=cin, cout=cout, M=M)
            self.rec[name] = r
            return r
        # z, the conv output BatchNorm normalises: written once, read three times (normalise, the two backward passes) and never an MFMA operand.  On
        # the bf16 path it is kept in fp16 -- the type the reference's autocast gives it -- with the batch statistics taken from the rounded values
        z16 = self.training and self.dt == torch.bfloat16 and not _FP32_Z and not _SEPARATE_STATS
        z = _e((M, cout), dev, torch.float16 if z16 else torch.float32)
        st = _z((2, STAT_COPIES, cout), dev) if self.training else (None, None)   # batch statistics ride on the conv's epilogue
        if _SEPARATE_STATS and self.training:
            ops.gemm_nt(xin, S.extra[p + ".0.weight::K"], z, M, cout, 9 * cin, ld_in, 9 * cin, cout, a_map=amap)
            ops.col_stats(z, cout, M, cout, st[0][0], st[1][0])
        else:
            ops.gemm_nt(xin, S.extra[p + ".0.weight::K"], z, M, cout, 9 * cin, ld_in, 9 * cin, cout, a_map=amap, col_sum=st[0], col_sumsq=st[1],
                        col_copies=STAT_COPIES)
        fin = None
        if self.training:
            mean, rstd = _e((cout,), dev), _e((cout,), dev)
            if z16 and cout % 8 == 0 and cout <= 256 and not _NO_FIN_FUSE:
                # the statistics are finalised in the prologue of the normalisation launch (`norm`): eleven tiny launches fewer per step
                fin = (st[0], st[1], bn.running_mean, bn.running_var)
            else:
                ops.bn_finalize(st[0], st[1], M, cout, BN_EPS, BN_MOM, mean, rstd, bn.running_mean, bn.running_var, copies=STAT_COPIES)
            self.nbt.append(bn.num_batches_tracked)        # all eleven counters are bumped by one launch at the end of forward
        else:
            mean = bn.running_mean
            rstd = torch.rsqrt(bn.running_var + BN_EPS)
        r = dict(name=name, p=p, z=z, mean=mean, rstd=rstd, xin=xin, ld_in=ld_in, amap=amap, cin=cin, cout=cout, M=M, side=side,
                 tokens_in=tokens_in, fin=fin)
        self.rec[name] = r
        return r

    def _folded(self, name, p, bn, cin, cout):
        """-> (W', b'): W' = W * gamma * rsqrt(running_var + eps) per output channel, taken from the fp32 masters in the gather's
        [out][kh][kw][cin] order and rounded to the operand dtype ONCE; b' = beta - running_mean * gamma * rsqrt(...).  Cached on the model
        until the parameters' derived copies are refreshed again or the BatchNorm buffers are written."""
        S, m = self.S, self.m
        key = (S.refresh_count, bn.running_mean._version, bn.running_var._version, bn.num_batches_tracked._version, self.dt)
        cache = m.__dict__.setdefault("_mim_fold", {})
        hit = cache.get(name)
        if hit is None or hit[0] != key:
            scale = S.master(p + ".1.weight") * torch.rsqrt(bn.running_var.float() + BN_EPS)
            shift = (S.master(p + ".1.bias") - bn.running_mean.float() * scale).contiguous()
            wk = (S.master(p + ".0.weight").permute(0, 2, 3, 1).reshape(cout, 9 * cin) * scale[:, None]).to(self.dt).contiguous()
            hit = cache[name] = (key, wk, shift)
        return hit[1], hit[2]

    def norm(self, r, y32=None, ld32=0, y16=None, ld16=0):
        S = self.S
        if "fold" in r:
            wk, shift = r["fold"]
            for y, ld in ((y32, ld32), (y16, ld16)):
                if y is not None:
                    ops.gemm_nt(r["xin"], wk, y, r["M"], r["cout"], 9 * r["cin"], r["ld_in"], 9 * r["cin"], ld, a_map=r["amap"], bias=shift)
            return
        if r.get("fin") is not None:
            s1, s2, rm, rv = r["fin"]
            r["fin"] = None                                # (a second `norm` of the same record must not update the running statistics again)
            ops.bn_finalize_norm(r["z"], r["cout"], s1, s2, STAT_COPIES, BN_EPS, BN_MOM, r["mean"], r["rstd"], rm, rv, S.master(r["p"] + ".1.weight"),
                                 S.master(r["p"] + ".1.bias"), r["M"], r["cout"], y32, ld32, y16, ld16)
            return
        ops.bn_norm(r["z"], r["cout"], r["mean"], r["rstd"], S.master(r["p"] + ".1.weight"), S.master(r["p"] + ".1.bias"), r["M"], r["cout"],
                    y32, ld32, y16, ld16)

    def up2(self, x32, ldx, side, C, out=None, ldo=None):
        B = self.B
        if out is None:
            out, ldo = _e((B * 4 * side * side, C), self.dev, self.dt), C
        ops.upsample_fwd(x32, ldx, B, side, side, C, 2, out, ldo)
        return out

    # ------------------------------------------------------------------ forward
    def forward(self, target=None):
        S, dev, dt, B = self.S, self.dev, self.dt, self.B
        x2, x3, x4 = self.x
        s1, s2, s3, M1, M2, M3 = self.s1, self.s2, self.s3, self.M1, self.M2, self.M3
        C2, C3, C4 = x2.shape[2], x3.shape[2], x4.shape[2]
        ch = 64
        f = self.rec
        # reductions of the three pyramid levels to 64 channels
        # the three factors of the full-resolution feature product (low, cu2o, cu3o: written once, read by the product and by its backward, never an
        # MFMA operand) are fp16 beside the fp16 z on the bf16 training path: 0.3 GB less traffic per step
        f16 = torch.float16 if (self.training and dt == torch.bfloat16 and not _FP32_Z and not _SEPARATE_STATS and not _FP32_DY) else torch.float32
        r = self.conv_bn("reduction1", x2, C2, x2.shape[1], s1, C2, ch, M1); low = _e((M1, ch), dev, f16); self.norm(r, low, ch)
        r = self.conv_bn("reduction2", x3, C3, x3.shape[1], s2, C3, ch, M2); mid = _e((M2, ch), dev); self.norm(r, mid, ch)
        r = self.conv_bn("reduction3", x4, C4, x4.shape[1], s3, C4, ch, M3); high = _e((M3, ch), dev); self.norm(r, high, ch)
        uph = self.up2(high, ch, s3, ch)                                   # (B,16,16,64) operand dtype
        # a = cu1(up(high)) * mid            -> fp32 + operand copy into cat2[:, :64]
        cat2 = _e((M2, 2 * ch), dev, dt)
        r = self.conv_bn("conv_upsample1", uph, ch, s2 * s2, s2, ch, ch, M2); cu1o = _e((M2, ch), dev); self.norm(r, cu1o, ch)
        a = _e((M2, ch), dev)
        ops.ew_mul(a, ch, cu1o, ch, mid, ch, M=M2, Cdim=ch, out16=cat2, ld16=2 * ch)
        r = self.conv_bn("conv_upsample4", uph, ch, s2 * s2, s2, ch, ch, M2); self.norm(r, y16=cat2[:, ch:], ld16=2 * ch)
        r = self.conv_bn("conv_concat2", cat2, 2 * ch, s2 * s2, s2, 2 * ch, 2 * ch, M2); c = _e((M2, 2 * ch), dev); self.norm(r, c, 2 * ch)
        # b = cu2(up(mid)) * cu3(up(a)) * low -> fp32 + operand copy into cat3[:, :64]
        cat3 = _e((M1, 3 * ch), dev, dt)
        upm = self.up2(mid, ch, s2, ch)
        r = self.conv_bn("conv_upsample2", upm, ch, s1 * s1, s1, ch, ch, M1); cu2o = _e((M1, ch), dev, f16); self.norm(r, cu2o, ch)
        upa = self.up2(a, ch, s2, ch)
        r = self.conv_bn("conv_upsample3", upa, ch, s1 * s1, s1, ch, ch, M1); cu3o = _e((M1, ch), dev, f16); self.norm(r, cu3o, ch)
        ops.ew_mul(None, 0, cu2o, ch, cu3o, ch, low, ch, M=M1, Cdim=ch, out16=cat3, ld16=3 * ch)
        upc = self.up2(c, 2 * ch, s2, 2 * ch)
        r = self.conv_bn("conv_upsample5", upc, 2 * ch, s1 * s1, s1, 2 * ch, 2 * ch, M1); self.norm(r, y16=cat3[:, ch:], ld16=3 * ch)
        d16 = _e((M1, 3 * ch), dev, dt)
        r = self.conv_bn("conv_concat3", cat3, 3 * ch, s1 * s1, s1, 3 * ch, 3 * ch, M1); self.norm(r, y16=d16, ld16=3 * ch)
        e16 = _e((M1, 3 * ch), dev, dt)
        r = self.conv_bn("conv4", d16, 3 * ch, s1 * s1, s1, 3 * ch, 3 * ch, M1); self.norm(r, y16=e16, ld16=3 * ch)
        # score: conv1x1 (192 -> 3) + bias, then x8 bilinear to the image, written as NCHW fp32
        sc = _e((M1, 3), dev)
        ops.gemm_nt(e16, S.extra["t2i_head.score.0.weight::W"], sc, M1, 3, 3 * ch, 3 * ch, 3 * ch, 3, bias=S.master("t2i_head.score.0.bias"))
        if target is not None:
            # training with the loss fused behind the decoder: SmoothL1 against the target image while interpolating, the (B, 3, S, S)
            # prediction is never written (and the backward recomputes it from the score map)
            acc = pool_zeros((1,), torch.float32, dev)
            ops.upsample_l1_fwd(sc, 3, B, s1, s1, 3, 8, target, acc)
            out = (acc / target.numel()).reshape(())
            self.sc, self.target = sc, target
        else:
            out = _e((B, 3, 8 * s1, 8 * s1), dev)
            ops.upsample_fwd(sc, 3, B, s1, s1, 3, 8, out, 0, nchw=True)
        if self.nbt:
            torch._foreach_add_(self.nbt, 1)
            self.nbt = []
        if self.need_grad:
            self.keep = dict(low=low, mid=mid, cu1o=cu1o, cu2o=cu2o, cu3o=cu3o, e16=e16)
        else:
            self.rec = {}
        return out

    # ------------------------------------------------------------------ backward
    def bn_conv_bwd(self, name, dy, lddy, dx=None, lddx=0, accumulate=False, dx_map=None, dx_dtype=torch.float32):
        """dy: fp32 gradient w.r.t. the BN output [M, cout] (row stride lddy).  Accumulates the BN / conv parameter gradients
        into the flat buffer and returns (or accumulates into) the gradient w.r.t. the conv input."""
        S, dev, dt = self.S, self.dev, self.dt
        r = self.rec[name]
        p, M, cin, cout = r["p"], r["M"], r["cin"], r["cout"]
        red = _z((2, cout), dev)
        ops.bn_bwd_reduce(dy, lddy, r["z"], cout, r["mean"], r["rstd"], M, cout, red[0], red[1])
        dz = _e((M, cout), dev, dt)
        ops.bn_bwd_apply(dy, lddy, r["z"], cout, r["mean"], r["rstd"], S.master(p + ".1.weight"), red[0], red[1], M, cout, dz, cout,
                         g_beta=S.grad(p + ".1.bias"), g_gamma=S.grad(p + ".1.weight"))
        # wgrad computed in the gather's [out][dy][dx][cin] order, accumulated at nn.Conv2d's [out][cin][3][3] place
        from .schedule import conv_wgrad
        conv_wgrad(S, p + ".0.weight", dz, r["xin"], M, cout, 9 * cin, cout, r["ld_in"], r["amap"], 9, cin)
        # dgrad: gather dz over the same grid with flipped taps
        gmap = conv3map(r["side"], r["side"], r["side"] * r["side"], cout)
        if dx is None:
            dx, lddx = _e((M, cin), dev, dx_dtype), cin
        ops.gemm_nt(dz, S.extra[p + ".0.weight::F"], dx, M, cin, 9 * cout, cout, 9 * cout, lddx, a_map=gmap, c_map=dx_map,
                    R=dx if accumulate else None)
        return dx

    def backward(self, dout, sink=None):
        S, dev, dt, B = self.S, self.dev, self.dt, self.B
        s1, s2, s3, M1, M2, M3 = self.s1, self.s2, self.s3, self.M1, self.M2, self.M3
        ch = 64
        k = self.keep
        # score head
        dsc_p = _z((M1, 8), dev, dt)                     # [pixels][3 -> 8] in the compute dtype: the operand of the two GEMMs below
        if getattr(self, "target", None) is not None:    # fused loss: dout is the scalar gradient of the loss
            ops.upsample_l1_bwd(self.sc, 3, B, s1, s1, 3, 8, self.target, dout.reshape(1).float().contiguous(), dsc_p, 8)
            self.sc = self.target = None
        else:
            dout = dout.contiguous().float()
            ops.upsample_bwd(dout, 0, True, B, s1, s1, 3, 8, dsc_p, 8)
        # weight gradient and, as the GEMM's column sum, the bias gradient (a torch sum over a [262144, 3] matrix took 92 us)
        ops.gemm_tn(dsc_p, k["e16"], S.grad("t2i_head.score.0.weight").view(3, 3 * ch), M1, 3, 3 * ch, 8, 3 * ch, 3 * ch,
                    colsum=S.grad("t2i_head.score.0.bias"))
        # the three 192-channel gradient maps at full resolution travel in the operand dtype (each is written once and read twice by the
        # BatchNorm backward behind it: 0.9 GB less traffic per step at batch 256); the maps further down stay fp32 (they are accumulated into)
        gd = dt if (dt == torch.bfloat16 and not _FP32_DY) else torch.float32
        de = _e((M1, 3 * ch), dev, gd)
        ops.gemm_nt(dsc_p, S.extra["t2i_head.score.0.weight::T"], de, M1, 3 * ch, 8, 8, 8, 3 * ch)
        dd = self.bn_conv_bwd("conv4", de, 3 * ch, dx_dtype=gd)
        dcat3 = self.bn_conv_bwd("conv_concat3", dd, 3 * ch, dx_dtype=gd)          # [:, :64] = db, [:, 64:] = d(cu5 out)
        dupc = self.bn_conv_bwd("conv_upsample5", dcat3[:, ch:], 3 * ch, dx_dtype=gd)
        dc = _e((M2, 2 * ch), dev)
        ops.upsample_bwd(dupc, 2 * ch, False, B, s2, s2, 2 * ch, 2, dc, 2 * ch)
        # b = cu2o * cu3o * low
        db = dcat3                                                                  # columns [0, 64), row stride 192
        dcu2o, dcu3o, dlow = _e((M1, ch), dev, gd), _e((M1, ch), dev, gd), _e((M1, ch), dev, gd)
        ops.ew_mul3_bwd(db, 3 * ch, k["cu2o"], k["cu3o"], k["low"], ch, dcu2o, dcu3o, dlow, M1, ch)
        # gradient of cat2 = [a | cu4 out]: starts with the path a -> up -> cu3
        dcat2 = _z((M2, 2 * ch), dev)
        dupa = self.bn_conv_bwd("conv_upsample3", dcu3o, ch, dx_dtype=gd)
        ops.upsample_bwd(dupa, ch, False, B, s2, s2, ch, 2, dcat2, 2 * ch, accumulate=True)
        dupm = self.bn_conv_bwd("conv_upsample2", dcu2o, ch, dx_dtype=gd)
        dmid = _e((M2, ch), dev)
        ops.upsample_bwd(dupm, ch, False, B, s2, s2, ch, 2, dmid, ch)
        self.bn_conv_bwd("conv_concat2", dc, 2 * ch, dx=dcat2, lddx=2 * ch, accumulate=True)
        duph = self.bn_conv_bwd("conv_upsample4", dcat2[:, ch:], 2 * ch)
        # a = cu1o * mid
        dcu1o = _e((M2, ch), dev)
        ops.ew_mul(dcu1o, ch, dcat2, 2 * ch, k["mid"], ch, M=M2, Cdim=ch)
        ops.ew_mul(dmid, ch, dcat2, 2 * ch, k["cu1o"], ch, M=M2, Cdim=ch, accumulate=True)
        self.bn_conv_bwd("conv_upsample1", dcu1o, ch, dx=duph, lddx=ch, accumulate=True)
        dhigh = _e((M3, ch), dev)
        ops.upsample_bwd(duph, ch, False, B, s3, s3, ch, 2, dhigh, ch)
        # reductions: gradients w.r.t. the image tokens of the stage outputs (text rows stay zero)
        grads = []
        for name, dy, x, side in (("reduction1", dlow, self.x[0], s1), ("reduction2", dmid, self.x[1], s2), ("reduction3", dhigh, self.x[2], s3)):
            if name == "reduction3" and sink is not None:
                dxs, ret = sink.take(x.shape, x.dtype, x.device)      # the heads' common buffer: image rows are this decoder's
            else:
                dxs = ret = S.own(torch.empty_like(x))     # image rows are all written by the conv dgrad below;
                dxs[:, side * side:].zero_()               # only the text rows need the explicit zeros
            self.bn_conv_bwd(name, dy, ch, dx=dxs, lddx=x.shape[2], dx_map=rowmap(side * side, x.shape[1], 0))
            grads.append(ret)
        self.rec, self.keep = {}, {}
        return grads


class _MimFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2, x3, x4, model, sides, need_grad, sink=None, target=None):
        step = MimStep(model, x2, x3, x4, sides, model.training, need_grad)
        out = step.forward(target)
        ctx.step, ctx.sink = step, sink
        return out

    @staticmethod
    def backward(ctx, dout):
        step = ctx.step
        step.S.queue_finalize()
        g2, g3, g4 = step.backward(dout, ctx.sink)
        ctx.step = ctx.sink = None
        step.S.fold_copies(early=True)                # the conv weight gradients leave the tap arena for G (with a data-parallel wrapper: now)
        step.S.announce_prefix("t2i_head.")          # the decoder's gradients are final: reduce them under the trunk backward
        return g2, g3, g4, None, None, None, None, None


def mim_head(model, x2, x3, x4, sides, need_grad, sink=None, target=None):
    """-> t2i_logits (B, 3, S, S), or with `target` (fp32 NCHW image) the mean SmoothL1 loss against it as a 0-dim tensor"""
    return _MimFn.apply(x2, x3, x4, model, sides, need_grad, sink, target)
