"""Forward / backward kernel schedules of the MI355X PVLT (the work that reference libs/pvlt.py:322-401 and its
autograd graph do with dozens of ATen kernels per block).

Data layout: one token-major activation buffer (B, HW+T, C) per stage in the compute dtype; image tokens first.
  * PatchEmbed / Attention.sr (kernel==stride convs) read it through a patch row-map inside the GEMM loader
  * text tokens are the row range [HW, HW+T) (row-map), so there is no torch.cat / torch.split / NCHW permute
  * LN(+pos-embed) epilogues write straight into the concatenated buffer
Backward is scheduled by hand: every weight gradient is accumulated by the wgrad GEMM directly into the flat fp32
gradient buffer (no autograd accumulate pass), input gradients reuse buffers in place.
"""
import os

import torch
import torch.nn.functional as F

from . import ops
from ._lib import patchmap, rowmap
from .params import ZeroPool, pool_zeros
from .pvlt import BERT_DROP, EPS_BERT, EPS_BLOCK, EPS_DEFAULT, VOCAB, VOCAB_LD


def _empty(shape, dtype, dev):
    return torch.empty(shape, dtype=dtype, device=dev)


_LN_COPIES = not os.environ.get("MVLT_LN_NO_COPIES")      # A/B switch: LayerNorm parameter gradients by plain atomics
_NO_DX2 = bool(os.environ.get("MVLT_NO_DX2"))      # A/B switch: DropPath-scaled gradient copy by a separate pass
_NO_T2I_FUSE = bool(os.environ.get("MVLT_NO_T2I_FUSE"))    # A/B switch: t2i_logits materialised, SmoothL1 as its own two passes
_NO_LN_CHAIN = bool(os.environ.get("MVLT_NO_LN_CHAIN")) or bool(os.environ.get("MVLT_LN_GENERIC"))   # A/B switch: first block's norm1 as its own launch
_NO_LNB_FUSE = bool(os.environ.get("MVLT_NO_LNB_FUSE"))    # A/B switch: norm2's backward as its own launch behind the fused-MLP dx kernel
_NO_POST_LN = bool(os.environ.get("MVLT_NO_POST_LN"))      # A/B switch: every block launches its own norm1
_NO_OUT_OP = bool(os.environ.get("MVLT_NO_OUT_OP"))        # A/B switch: fp32 stage output + separate cast pass
_NO_PROJ_LN = bool(os.environ.get("MVLT_NO_PROJ_LN"))      # A/B switch: LN2 folded into the fused MLP's operand load (round 2) instead of the proj epilogue
_NO_LN_FOLD = bool(os.environ.get("MVLT_NO_LN_FOLD"))      # A/B switch: LN2 as its own launch in front of the fused MLP
_NO_POS_BATCH = bool(os.environ.get("MVLT_NO_POS_BATCH"))  # A/B switch: one resize launch per stage and direction for the position embeddings
_NT_GENERIC_EPI = bool(os.environ.get("MVLT_NT_GENERIC_EPI"))   # the library's switch (generic GEMM epilogue): the r_fp32 stage output does not exist there
_NO_LIN_FUSE = bool(os.environ.get("MVLT_NO_LIN_FUSE"))    # A/B switch: weight and input gradient of the C x C Linears of stages 1-2 as two launches
# A/B switches: conv weight gradients accumulated straight into the [out][cin][kh][kw] layout by the wgrad epilogue (strided atomics),
# or through a pooled [out][kh][kw][cin] buffer + one permuted ATen add per convolution, instead of the store's tap arena
WGRAD_TAPS = bool(os.environ.get("MVLT_WGRAD_TAPS"))
WGRAD_ADD = bool(os.environ.get("MVLT_WGRAD_ADD"))


_NO_OVERWRITE = bool(os.environ.get("MVLT_TN_NO_OVERWRITE"))     # 1: the vocabulary decoder's weight gradient back on fp32 atomics (round 5)
_DEFER_FOLD = os.environ.get("MVLT_TN_DEFER_FOLD", "1") != "0"     # the weight-gradient folds batched into a few launches (FlatStore.fold_copies flushes); 0 = one fold per GEMM


def conv_wgrad(S, name, dz, xin, M, cout, K, ld_dz, ld_in, bmap, taps, cin, colsum=None):
    """weight gradient of a gathered-row convolution into the G slice of nn.Conv2d's (out, cin, kh, kw) weight `name`: accumulated in
    the gather's (out, kh, kw, cin) order in the store's tap arena, which S.fold_copies() adds to G at the (out, cin, kh, kw) places"""
    if WGRAD_TAPS:
        ops.gemm_tn(dz, xin, S.grad(name).view(cout, K), M, cout, K, ld_dz, ld_in, K, b_map=bmap, colsum=colsum, taps=taps, seg=cin)
        return
    if WGRAD_ADD:
        dWk = pool_zeros((cout, K), torch.float32, dz.device)
        ops.gemm_tn(dz, xin, dWk, M, cout, K, ld_dz, ld_in, K, b_map=bmap, colsum=colsum)
        kk = int(round(taps ** 0.5))
        S.grad(name).add_(dWk.view(cout, kk, kk, cin).permute(0, 3, 1, 2))
        return
    ops.gemm_tn(dz, xin, S.grad_taps(name, cout, taps, cin), M, cout, K, ld_dz, ld_in, K, b_map=bmap, colsum=colsum, partials=S.tn_partials(), defer_fold=_DEFER_FOLD)


def _step_rng(model):
    """(seed, call) of the next draw of the step's own train-mode masks (BERT dropout, DropPath): the seed is torch's
    (torch.manual_seed(args.seed + rank), reference main_vl.py:207-209, so ranks draw different masks), the counter runs per model"""
    model._rng_calls = getattr(model, "_rng_calls", 0) + 1
    return torch.initial_seed(), model._rng_calls


class Names:
    """parameter-name helpers"""

    @staticmethod
    def blk(i, j):
        return f"block{i+1}.{j}."


# =============================================================================================== trunk
class TrunkStep:
    """One forward (and optionally backward) of the 4-stage trunk."""

    def __init__(self, model, images, ids, need_grad):
        self.m = model
        self.S = model.store
        self.dev = images.device
        self.dt = model.compute_dtype       # GEMM / attention operand dtype
        self.rt = torch.float32             # residual-stream dtype: adds and LayerNorm inputs stay fp32 (what the
        #                                     reference's autocast region does too); only MFMA operands are bf16
        self.need_grad = need_grad
        self.images = images.contiguous().float()
        self.ids = ids.contiguous()
        self.B = images.shape[0]
        assert images.shape[1] == model.in_chans
        Himg, Wimg = images.shape[2], images.shape[3]
        assert Himg == Wimg, "square inputs only (the reference's pos-embed handling assumes them)"
        assert Himg % (model.patch_size * 8) == 0, f"img_size {Himg} should be divided by patch_size {model.patch_size * 8}."
        self.img = Himg
        self.T = ids.shape[1]
        assert self.T == model.T_num, "input_ids length must equal num_text_tokens"
        self.side = [Himg // model.patch_size // (2 ** i) for i in range(4)]
        self.saved = []          # per stage dict
        self.training = model.training

    def _mark(self):
        """bench.py: HIP events around the Block kernels of a stage (SRAttention + MLP incl. their LayerNorms), in pairs, on the
        launch stream -- the blocks-only GPU time behind north_star's MFMA-utilisation figure.  Off unless model._block_events
        is a list."""
        ev = getattr(self.m, "_block_events", None)
        if ev is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            ev.append(e)

    # ---- parameter access
    def w(self, name):
        return self.S.comp(name)

    def wT(self, name):
        return self.S.extra[name + "::T"]

    def wK(self, name):
        return self.S.extra[name + "::K"]

    def wKT(self, name):
        return self.S.extra[name + "::KT"]

    def f32(self, name):
        return self.S.master(name)

    def g(self, name):
        return self.S.grad(name)

    def gl(self, name):
        """LayerNorm weight / bias gradient: slot of the interleaved accumulators (FlatStore.grad_copies)"""
        return self.S.grad_copies(name) if _LN_COPIES else self.S.grad(name)

    def lnk(self):
        return self.S.ln_kwargs() if _LN_COPIES else {}

    # ---- pos embed: learned for the constructor grid, bilinearly resized (align_corners=False) to this input's grid
    def _pos(self, i, param):
        m = self.m
        side, C = self.side[i], m.dims[i]
        HW = side * side
        pe = (param[:, 1:] if i == 3 else param)[0]                 # (grid*grid, C) rows of the flat buffer; stage 4 skips the cls slot
        if HW == m.grids[0] ** 2:          # reference libs/pvlt.py:292 compares with stage-1's constructor grid
            return pe
        pre = getattr(self, "_pos_pre", None)
        if pre is not None and i in pre:
            return pre[i]                      # resized by the one launch at the start of the forward pass (`_pos_prefetch`)
        out = _empty((HW, C), torch.float32, self.dev)
        ops.resize_bilinear_tokens(pe, out, m.grids[i], m.grids[i], side, side, C)
        return out

    def _pos_prefetch(self):
        """the resized position embeddings of all four stages in ONE launch (they depend on parameters only)"""
        m = self.m
        jobs, self._pos_pre = [], {}
        for i in range(4):
            side, C = self.side[i], m.dims[i]
            if side * side == m.grids[0] ** 2:
                continue
            param = self.f32(f"pos_embed{i+1}")
            pe = (param[:, 1:] if i == 3 else param)[0]
            out = _empty((side * side, C), torch.float32, self.dev)
            jobs.append((pe, out, m.grids[i], m.grids[i], side, side, C))
            self._pos_pre[i] = out
        if jobs and not _NO_POS_BATCH:
            ops.resize_bilinear_tokens_multi(jobs)
        else:
            self._pos_pre = None

    def _droppath_scales(self, blk_index):
        m = self.m
        rate = m.dpr[blk_index]
        if not self.training or rate == 0.0:
            return None, None
        inj = m.injected_masks
        if inj is not None:
            k1 = inj["droppath"][blk_index].to(self.dev, torch.float32)
            k2 = inj["droppath2"][blk_index].to(self.dev, torch.float32)
        else:
            if getattr(self, "_dp_all", None) is None:
                # every block's two keep masks in one draw (4 small launches per step instead of 8 per block)
                rates = getattr(m, "_dpr_dev", None)
                if rates is None or rates.device != self.dev:
                    rates = m._dpr_dev = torch.tensor(list(m.dpr), device=self.dev, dtype=torch.float32)
                self._dp_all = ops.droppath_scales(_empty((len(m.dpr), 2, self.B), torch.float32, self.dev), rates, *_step_rng(m))
            return self._dp_all[blk_index, 0], self._dp_all[blk_index, 1]
        return (k1 / (1.0 - rate)).contiguous(), (k2 / (1.0 - rate)).contiguous()

    # ------------------------------------------------------------------ forward
    def forward(self):
        m, S, B, T, dt, dev = self.m, self.S, self.B, self.T, self.dt, self.dev
        te = "text_embeddings."
        self._pos_prefetch()
        # BERT embeddings (+LN eps 1e-12, +dropout in train mode)
        rows = B * T
        self.emb = _empty((rows, m.hidden), dt, dev)
        self.emb_mean = _empty((rows,), torch.float32, dev)
        self.emb_rstd = _empty((rows,), torch.float32, dev)
        self.keep = None
        if self.training:
            inj = m.injected_masks
            if inj is not None:
                self.keep = inj["bert"].to(dev).reshape(rows, m.hidden).to(torch.uint8).contiguous()
            else:
                self.keep = ops.keep_mask(_empty((rows, m.hidden), torch.uint8, dev), BERT_DROP, *_step_rng(m))
        ops.bert_embed_fwd(self.ids, self.f32(te + "word_embeddings.weight"), self.f32(te + "position_embeddings.weight"),
                           self.f32(te + "token_type_embeddings.weight"), self.f32(te + "LayerNorm.weight"),
                           self.f32(te + "LayerNorm.bias"), self.keep, BERT_DROP, self.emb, self.emb_mean, self.emb_rstd,
                           rows, T, EPS_BERT)
        xp = None
        blk_index = 0
        outs = []
        for i in range(4):
            xp, blk_index = self._stage_forward(i, xp, blk_index)
            outs.append(xp)
        return outs

    def _stage_forward(self, i, xp, blk_index):
        m, B, T, dt, dev = self.m, self.B, self.T, self.dt, self.dev
        C = m.dims[i]
        side = self.side[i]
        HW = side * side
        N = HW + T
        sv = dict(i=i, C=C, HW=HW, N=N, side=side)
        pe, ten = f"patch_embed{i+1}.", f"text_embed{i+1}."
        # ---- patch embed: kernel==stride conv as GEMM, then LN(1e-5) + pos-embed written into x[:, :HW]
        pe_pre = _empty((B * HW, C), dt, dev)
        if i == 0:
            K = m.in_chans * m.patch_size ** 2
            P1 = _empty((B * HW, K), dt, dev)
            ops.patchify(self.images, P1, B, m.in_chans, self.img, self.img, m.patch_size)
            ops.gemm_nt(P1, self.w(pe + "proj.weight"), pe_pre, B * HW, C, K, K, K, C, bias=self.f32(pe + "proj.bias"))
            sv["P1"] = P1
        else:
            Cp, Np, sp = m.dims[i - 1], self.saved[i - 1]["N"], self.side[i - 1]
            pm = patchmap(2, sp, Np, HW, side, Cp)
            ops.gemm_nt(xp, self.wK(pe + "proj.weight"), pe_pre, B * HW, C, 4 * Cp, Cp, 4 * Cp, C, a_map=pm,
                        bias=self.f32(pe + "proj.bias"))
            sv["pm_in"] = pm
        x = _empty((B, N, C), self.rt, dev)
        pos = self._pos(i, self.f32(f"pos_embed{i+1}"))
        sv["pe_pre"], sv["pe_mean"], sv["pe_rstd"] = pe_pre, _empty((B * HW,), torch.float32, dev), _empty((B * HW,), torch.float32, dev)
        # the first block's norm1 is chained onto the two embedding LayerNorms below (their output rows are in registers): that block
        # launches no LayerNorm of its own and the fp32 rows are not read back
        chain = None
        if dt == torch.bfloat16 and C in ops.LN_CHAIN_WIDTHS and not _NO_LN_CHAIN:
            p0 = Names.blk(i, 0)
            self._pre_ln1 = (_empty((B, N, C), dt, dev), _empty((B * N,), torch.float32, dev), _empty((B * N,), torch.float32, dev))
            chain = (self.f32(p0 + "norm1.weight"), self.f32(p0 + "norm1.bias"), EPS_BLOCK, *self._pre_ln1)
        ops.layernorm_fwd(pe_pre, x, self.f32(pe + "norm.weight"), self.f32(pe + "norm.bias"), B * HW, C, C, C, EPS_DEFAULT,
                          mean=sv["pe_mean"], rstd=sv["pe_rstd"], add=pos, add_rows=HW, y_map=rowmap(HW, N, 0), chain=chain)
        # ---- text embed: Linear + LN(1e-5) + text pos-embed written into x[:, HW:]
        te_pre = _empty((B * T, C), dt, dev)
        if i == 0:
            ops.gemm_nt(self.emb, self.w(ten + "0.weight"), te_pre, B * T, C, m.hidden, m.hidden, m.hidden, C,
                        bias=self.f32(ten + "0.bias"))
        else:
            Cp, Np, HWp = m.dims[i - 1], self.saved[i - 1]["N"], self.saved[i - 1]["HW"]
            ops.gemm_nt(xp, self.w(ten + "0.weight"), te_pre, B * T, C, Cp, Cp, Cp, C, a_map=rowmap(T, Np, HWp),
                        bias=self.f32(ten + "0.bias"))
        sv["te_pre"], sv["te_mean"], sv["te_rstd"] = te_pre, _empty((B * T,), torch.float32, dev), _empty((B * T,), torch.float32, dev)
        ops.layernorm_fwd(te_pre, x, self.f32(ten + "1.weight"), self.f32(ten + "1.bias"), B * T, C, C, C, EPS_DEFAULT,
                          mean=sv["te_mean"], rstd=sv["te_rstd"], add=self.f32(f"text_pos_embed{i+1}")[0], add_rows=T,
                          y_map=rowmap(T, N, HW), chain=chain)
        sv["x_in_prev"] = xp
        sv["blocks"] = []
        self._mark()
        for j in range(m.depths[i]):
            x, bsv = self._block_forward(i, j, x, blk_index)
            sv["blocks"].append(bsv)
            blk_index += 1
        self._mark()
        if x.dtype != dt:                    # MFMA-operand copy of the stage output (next stage's convs, the heads)
            xb = _empty((B, N, C), dt, dev)
            ops.cast_bf16(x, xb, x.numel())
            x = xb
        sv["x_out"] = x
        taps = getattr(m, "_taps", None)
        if taps is not None:            # tests: stage outputs in the reference's (img_feat NCHW, text_feat) form
            taps[f"img_feat{i+1}"] = x[:, :HW].float().reshape(B, side, side, C).permute(0, 3, 1, 2)
            taps[f"text_feat{i+1}"] = x[:, HW:].float()
        self.saved.append(sv)
        return x, blk_index

    def _block_forward(self, i, j, x, blk_index):
        m, B, T, dt, dev = self.m, self.B, self.T, self.dt, self.dev
        C, h, r, hid = m.dims[i], m.heads[i], m.sr[i], m.hid[i]
        side = self.side[i]
        HW = side * side
        N = HW + T
        M = B * N
        p = Names.blk(i, j)
        f32 = torch.float32
        bs = dict(x=x)
        s1, s2 = self._droppath_scales(blk_index)
        bs["s1"], bs["s2"] = s1, s2
        # LN1 (already done by the previous block's fused MLP epilogue where there is one)
        pre = getattr(self, "_pre_ln1", None)
        self._pre_ln1 = None
        if pre is not None:
            xn1, bs["m1"], bs["r1"] = pre
        else:
            xn1 = _empty((B, N, C), dt, dev)
            bs["m1"], bs["r1"] = _empty((M,), f32, dev), _empty((M,), f32, dev)
            ops.layernorm_fwd(x, xn1, self.f32(p + "norm1.weight"), self.f32(p + "norm1.bias"), M, C, C, C, EPS_BLOCK, mean=bs["m1"], rstd=bs["r1"])
        bs["xn1"] = xn1
        # q
        q = _empty((B, N, C), dt, dev)
        ops.gemm_nt(xn1, self.w(p + "attn.q.weight"), q, M, C, C, C, C, C, bias=self.f32(p + "attn.q.bias"))
        bs["q"] = q
        # k, v source: spatially reduced image tokens (conv r x r stride r + LN 1e-5) followed by the text tokens
        if r > 1:
            sr_side = side // r
            HWr = sr_side * sr_side
            Mk = HWr + T
            pm = patchmap(r, side, N, HWr, sr_side, C)
            sr_pre = _empty((B * HWr, C), dt, dev)
            ops.gemm_nt(xn1, self.wK(p + "attn.sr.weight"), sr_pre, B * HWr, C, r * r * C, C, r * r * C, C, a_map=pm,
                        bias=self.f32(p + "attn.sr.bias"))
            kvin = _empty((B * HWr, C), dt, dev)
            bs["msr"], bs["rsr"] = _empty((B * HWr,), f32, dev), _empty((B * HWr,), f32, dev)
            ops.layernorm_fwd(sr_pre, kvin, self.f32(p + "attn.norm.weight"), self.f32(p + "attn.norm.bias"), B * HWr, C, C, C,
                              EPS_DEFAULT, mean=bs["msr"], rstd=bs["rsr"])
            kv = _empty((B, Mk, 2 * C), dt, dev)
            wkv, bkv = self.w(p + "attn.kv.weight"), self.f32(p + "attn.kv.bias")
            ops.gemm_nt(kvin, wkv, kv, B * HWr, 2 * C, C, C, C, 2 * C, c_map=rowmap(HWr, Mk, 0), bias=bkv)
            ops.gemm_nt(xn1, wkv, kv, B * T, 2 * C, C, C, C, 2 * C, a_map=rowmap(T, N, HW), c_map=rowmap(T, Mk, HWr), bias=bkv)
            bs.update(sr_pre=sr_pre, kvin=kvin, pm=pm, HWr=HWr)
        else:
            Mk = N
            kv = _empty((B, Mk, 2 * C), dt, dev)
            ops.gemm_nt(xn1, self.w(p + "attn.kv.weight"), kv, M, 2 * C, C, C, C, 2 * C, bias=self.f32(p + "attn.kv.bias"))
        bs["kv"], bs["Mk"] = kv, Mk
        # attention core
        ao = _empty((B, N, C), dt, dev)
        lse = _empty((B, h, N), f32, dev)
        ops.sr_attention_fwd(q, kv, ao, lse, B, h, N, Mk, C, 2 * C, C, 0, C, 64 ** -0.5)
        bs["ao"], bs["lse"] = ao, lse
        # proj + DropPath + residual
        xm = _empty((B, N, C), self.rt, dev)
        xn2 = _empty((B, N, C), dt, dev)
        bs["m2"], bs["r2"] = _empty((M,), f32, dev), _empty((M,), f32, dev)
        bs["xn2"] = xn2
        bs["fused_mlp"] = fused = (dt == torch.bfloat16 and C in (64, 128))
        # stages 1-2 (one tile of the projection holds whole rows): Block.norm2 rides on the projection's epilogue -- the fused MLP then reads
        # its operand in bf16 and the fp32 mid stream only once (as the residual), instead of normalising the fp32 rows itself
        proj_ln = fused and not _NO_PROJ_LN and not _NO_LN_FOLD and M < (1 << 24)      # (the lean GEMM epilogues index rows in 24 bits)
        ops.gemm_nt(ao, self.w(p + "attn.proj.weight"), xm, M, C, C, C, C, C, bias=self.f32(p + "attn.proj.bias"),
                    row_scale=s1, rows_per_scale=N, R=x,
                    post_ln=(self.f32(p + "norm2.weight"), self.f32(p + "norm2.bias"), EPS_BLOCK, xn2.view(M, C), bs["m2"], bs["r2"]) if proj_ln else None)
        bs["xm"] = xm
        # LN2 + MLP (fc1 + exact GELU, fc2) + DropPath + residual
        # the last block of a stage has no fp32 consumer (its output feeds the next stage's patch embedding and the heads, which read
        # the MFMA-operand copy): the fused MLP then writes that copy itself and no fp32 stream -- no separate cast pass
        last_op = fused and j == m.depths[i] - 1 and self.dt != self.rt and not _NO_OUT_OP
        # stages 3-4: the same through fc2's residual epilogue (bf16 C beside the fp32 residual, mvlt_gemm_nt_args.r_fp32)
        # (r_fp32 exists in the lean residual epilogue only: rows indexed in 24 bits, 16-byte row pieces -- otherwise the fp32 output + cast pass below)
        last_gemm = ((not fused) and j == m.depths[i] - 1 and dt == torch.bfloat16 and self.rt == torch.float32 and not _NO_OUT_OP and
                     M < (1 << 24) and C % 8 == 0 and not _NT_GENERIC_EPI)
        xo = _empty((B, N, C), dt if (last_op or last_gemm) else self.rt, dev)
        if fused:
            # stages 1-2: LN2 -> fc1 -> GELU -> fc2 -> DropPath -> +residual in ONE kernel.  The (tokens x hidden) activation stays on
            # chip and is recomputed by the fused backward kernels; LN2 is folded into the operand load (the kernel reads the fp32
            # mid stream once for both the normalisation and the residual, and stores LN2's output + statistics for the backward)
            ln = None if (_NO_LN_FOLD or proj_ln) else (self.f32(p + "norm2.weight"), self.f32(p + "norm2.bias"), EPS_BLOCK, xn2, bs["m2"], bs["r2"])
            if ln is None and not proj_ln:
                ops.layernorm_fwd(xm, xn2, self.f32(p + "norm2.weight"), self.f32(p + "norm2.bias"), M, C, C, C, EPS_BLOCK, mean=bs["m2"], rstd=bs["r2"])
            post = None
            if j + 1 < m.depths[i] and not _NO_POST_LN:
                # the next block's norm1 rides on this kernel's epilogue (the output row is in registers there)
                pn = Names.blk(i, j + 1)
                self._pre_ln1 = (_empty((B, N, C), dt, dev), _empty((M,), f32, dev), _empty((M,), f32, dev))
                post = (self.f32(pn + "norm1.weight"), self.f32(pn + "norm1.bias"), EPS_BLOCK, *self._pre_ln1)
            ops.mlp_fwd(None if ln else xn2, self.w(p + "mlp.fc1.weight"), self.f32(p + "mlp.fc1.bias"), self.w(p + "mlp.fc2.weight"),
                        self.f32(p + "mlp.fc2.bias"), xm, None if last_op else xo, M, C, hid, row_scale=s2, rows_per_scale=N, ln=ln,
                        out_op=xo if last_op else None, post_ln=post)
        else:
            ops.layernorm_fwd(xm, xn2, self.f32(p + "norm2.weight"), self.f32(p + "norm2.bias"), M, C, C, C, EPS_BLOCK, mean=bs["m2"], rstd=bs["r2"])
            hpre = _empty((M, hid), dt, dev) if self.need_grad else None
            gact = _empty((M, hid), dt, dev)
            ops.gemm_nt(xn2, self.w(p + "mlp.fc1.weight"), gact, M, hid, C, C, C, hid, bias=self.f32(p + "mlp.fc1.bias"), act=1, H=hpre)
            bs["hpre"], bs["gact"] = hpre, gact
            ops.gemm_nt(gact, self.w(p + "mlp.fc2.weight"), xo, M, C, hid, hid, hid, C, bias=self.f32(p + "mlp.fc2.bias"),
                        row_scale=s2, rows_per_scale=N, R=xm)
        if not self.need_grad:
            bs.clear()
        return xo, bs

    # ------------------------------------------------------------------ backward
    def _scaled(self, dy, scale, N):
        """dy * scale[b] per sample (DropPath); identity when scale is None."""
        if scale is None:
            return dy
        out = torch.empty_like(dy)
        ops.row_scale(dy, scale, N, self.B * N, dy.shape[-1], out)
        return out

    def backward(self, dxs):
        """dxs: gradients w.r.t. the four stage outputs (None allowed).  Fills the flat gradient buffer."""
        m, B, T, dt, dev, S = self.m, self.B, self.T, self.dt, self.dev, self.S
        self._pos_adj = []
        # gradient tensors the head nodes of this pass created themselves (the heads' shared buffer, the MIM decoder's outputs) may be
        # written in place; anything else autograd hands us is copied first
        own = [d is not None and d.dtype == dt and d.is_contiguous() and S.owns(d) for d in dxs]
        dx, merged = None, False
        for i in (3, 2, 1, 0):
            sv = self.saved[i]
            d_out = dxs[i]
            if d_out is not None and not merged:
                d_out = d_out.to(dt)
                if dx is None:
                    dx = d_out if own[i] else d_out.contiguous().clone()
                else:
                    dx.add_(d_out)
            if dx is None:
                continue
            # the stage's input gradient is accumulated straight into the previous stage's own head gradient where there is one
            into = dxs[i - 1] if i > 0 and own[i - 1] else None
            dx = self._stage_backward(i, sv, dx.view(B, sv["N"], sv["C"]), into)
            merged = into is not None
            self.S.announce_stage(i)
        # the learned position embeddings (root parameters: they sit in front of the stage blocks in the flat layout) got their last
        # contribution from stage 1's backward: final now, so that only the BERT embedding block is left for the end of the pass
        if self._pos_adj:
            ops.resize_bilinear_tokens_multi(self._pos_adj, adjoint=True)
            self._pos_adj = []
        self.S.announce_prefix("pos_embed", "text_pos_embed")
        self.S.fold_copies()
        self.saved = []

    def _stage_backward(self, i, sv, dx, into=None):
        m, B, T, dt, dev = self.m, self.B, self.T, self.dt, self.dev
        C, HW, N, side = sv["C"], sv["HW"], sv["N"], sv["side"]
        self._mark()
        for j in reversed(range(m.depths[i])):
            dx = self._block_backward(i, j, sv["blocks"][j], dx)
        self._mark()
        pe, ten = f"patch_embed{i+1}.", f"text_embed{i+1}."
        f32 = torch.float32
        # pos-embed / text-pos-embed gradients: sum over the batch of d(x0)
        dpos_all = _empty((N, C), f32, dev)
        ops.batch_sum(dx, dpos_all, B, N, C, N, C, acc2=self.g(f"text_pos_embed{i+1}")[0], split=HW)        # text rows straight into G
        self._pos_backward(i, dpos_all[:HW])
        # patch-embed LN backward -> d(pe_pre)
        d_pe = _empty((B * HW, C), dt, dev)
        ops.layernorm_bwd(dx, sv["pe_pre"], d_pe, self.f32(pe + "norm.weight"), sv["pe_mean"], sv["pe_rstd"], B * HW, C, C, C, C,
                          dgamma=self.gl(pe + "norm.weight"), dbeta=self.gl(pe + "norm.bias"), **self.lnk(), dy_map=rowmap(HW, N, 0))
        d_te = _empty((B * T, C), dt, dev)
        ops.layernorm_bwd(dx, sv["te_pre"], d_te, self.f32(ten + "1.weight"), sv["te_mean"], sv["te_rstd"], B * T, C, C, C, C,
                          dgamma=self.gl(ten + "1.weight"), dbeta=self.gl(ten + "1.bias"), **self.lnk(), dy_map=rowmap(T, N, HW))
        if i == 0:
            K = m.in_chans * m.patch_size ** 2
            ops.gemm_tn(d_pe, sv["P1"], self.g(pe + "proj.weight").view(C, K), B * HW, C, K, C, K, K, colsum=self.g(pe + "proj.bias"))
            # (round 6: the text-embedding weight gradients leave as partial tiles too -- 64 x 768 over 32768 rows: 64 splits' atomics on 1536 cache lines were 56 us)
            ops.gemm_tn(d_te, self.emb, self.g(ten + "0.weight"), B * T, C, m.hidden, C, m.hidden, m.hidden, colsum=self.g(ten + "0.bias"),
                        partials=self.S.tn_partials(), defer_fold=_DEFER_FOLD)
            d_emb = _empty((B * T, m.hidden), dt, dev)
            ops.gemm_nt(d_te, self.wT(ten + "0.weight"), d_emb, B * T, m.hidden, C, C, C, m.hidden)
            self._bert_backward(d_emb)
            return None
        Cp, Np, HWp, sp = m.dims[i - 1], self.saved[i - 1]["N"], self.saved[i - 1]["HW"], self.side[i - 1]
        xp = sv["x_in_prev"]
        pm = sv["pm_in"]
        # conv weight gradient: computed in the gather's [out][kh][kw][cin] order, accumulated at its [out][cin][kh][kw] place
        conv_wgrad(self.S, pe + "proj.weight", d_pe, xp, B * HW, C, 4 * Cp, C, Cp, pm, 4, Cp, colsum=self.g(pe + "proj.bias"))
        ops.gemm_tn(d_te, xp, self.g(ten + "0.weight"), B * T, C, Cp, C, Cp, Cp, b_map=rowmap(T, Np, HWp), colsum=self.g(ten + "0.bias"),
                    partials=self.S.tn_partials(), defer_fold=_DEFER_FOLD)
        dxp = into.view(B, Np, Cp) if into is not None else _empty((B, Np, Cp), dt, dev)
        ops.gemm_nt(d_pe, self.wKT(pe + "proj.weight"), dxp, B * HW, 4 * Cp, C, C, C, Cp, c_map=pm, R=into)          # image rows (each once)
        ops.gemm_nt(d_te, self.wT(ten + "0.weight"), dxp, B * T, Cp, C, C, C, Cp, c_map=rowmap(T, Np, HWp), R=into)   # text rows
        return dxp

    def _pos_backward(self, i, dpos):
        m = self.m
        side = self.side[i]
        HW = side * side
        gv = self.g(f"pos_embed{i+1}")
        gv = (gv[:, 1:] if i == 3 else gv)[0]
        if HW == m.grids[0] ** 2:
            gv.add_(dpos)
            return
        if _NO_POS_BATCH:
            ops.resize_bilinear_tokens(dpos, gv, m.grids[i], m.grids[i], side, side, dpos.shape[1], adjoint=True)   # accumulates into G
        else:
            # the adjoints of all stages leave in one launch at the end of the trunk's backward (`backward`): dpos stays alive in the list until then
            self._pos_adj.append((dpos, gv, m.grids[i], m.grids[i], side, side, dpos.shape[1]))

    def _bert_backward(self, d_emb):
        m, B, T = self.m, self.B, self.T
        te = "text_embeddings."
        ops.bert_embed_bwd(d_emb, self.ids, self.f32(te + "word_embeddings.weight"), self.f32(te + "position_embeddings.weight"),
                           self.f32(te + "token_type_embeddings.weight"), self.f32(te + "LayerNorm.weight"), self.keep, BERT_DROP,
                           self.emb_mean, self.emb_rstd, self.g(te + "word_embeddings.weight"), self.g(te + "position_embeddings.weight"),
                           self.g(te + "token_type_embeddings.weight"), self.g(te + "LayerNorm.weight"), self.g(te + "LayerNorm.bias"),
                           B * T, T)

    def _block_backward(self, i, j, bs, dx):
        """dx: gradient w.r.t. the block output (B,N,C), overwritten in place with the gradient w.r.t. its input."""
        m, B, T, dt, dev = self.m, self.B, self.T, self.dt, self.dev
        C, h, r, hid = m.dims[i], m.heads[i], m.sr[i], m.hid[i]
        side = self.side[i]
        HW = side * side
        N = HW + T
        M = B * N
        p = Names.blk(i, j)
        f32 = torch.float32
        # ---- MLP branch: x_out = x_mid + s2 * (fc2(gelu(fc1(LN2(x_mid)))))
        dxn2 = _empty((M, C), dt, dev)
        ln2_done = False
        if bs["fused_mlp"]:
            w1, w2t = self.w(p + "mlp.fc1.weight"), self.wT(p + "mlp.fc2.weight")
            ops.mlp_bwd_dw(bs["xn2"], dx, w1, w2t, self.f32(p + "mlp.fc1.bias"), self.g(p + "mlp.fc1.weight"), self.g(p + "mlp.fc1.bias"),
                           self.g(p + "mlp.fc2.weight"), self.g(p + "mlp.fc2.bias"), M, C, hid, row_scale=bs["s2"], rows_per_scale=N,
                           partials=self.S.tn_partials(), defer_fold=_DEFER_FOLD)
            if not _NO_LNB_FUSE:
                # ... and norm2's backward rides on the dx kernel's epilogue (the row of d(LN output) is in registers there): dx is
                # updated in place, the DropPath-scaled copy for the attention branch comes out of the same pass, no dxn2 round trip
                fuse = bs["s1"] is not None and not _NO_DX2
                dy1 = _empty((M, C), dx.dtype, dev) if fuse else dx
                ops.mlp_bwd_dx(bs["xn2"], dx, w1, self.wT(p + "mlp.fc1.weight"), w2t, self.f32(p + "mlp.fc1.bias"), None, M, C, hid,
                               row_scale=bs["s2"], rows_per_scale=N,
                               ln_bwd=dict(x=bs["xm"], mean=bs["m2"], rstd=bs["r2"], gamma=self.f32(p + "norm2.weight"), dx=dx,
                                           dgamma=self.gl(p + "norm2.weight"), dbeta=self.gl(p + "norm2.bias"),
                                           dx2=dy1 if fuse else None, dx2_scale=bs["s1"], dx2_rows_per_scale=N))
                ln2_done = True
            else:
                ops.mlp_bwd_dx(bs["xn2"], dx, w1, self.wT(p + "mlp.fc1.weight"), w2t, self.f32(p + "mlp.fc1.bias"), dxn2, M, C, hid,
                               row_scale=bs["s2"], rows_per_scale=N)
        else:
            dy2 = getattr(self, "_dy2_pre", None)                # written by the block above's norm1 backward when there is one
            self._dy2_pre = None
            if dy2 is None:
                dy2 = self._scaled(dx, bs["s2"], N)
            ops.gemm_tn(dy2, bs["gact"], self.g(p + "mlp.fc2.weight"), M, C, hid, C, hid, hid, colsum=self.g(p + "mlp.fc2.bias"), partials=self.S.tn_partials(), defer_fold=_DEFER_FOLD)
            dh = _empty((M, hid), dt, dev)
            ops.gemm_nt(dy2, self.wT(p + "mlp.fc2.weight"), dh, M, hid, C, C, C, hid, act=2, H=bs["hpre"])
            bs["gact"] = bs["hpre"] = None
            ops.gemm_tn(dh, bs["xn2"], self.g(p + "mlp.fc1.weight"), M, hid, C, hid, C, C, colsum=self.g(p + "mlp.fc1.bias"), partials=self.S.tn_partials(), defer_fold=_DEFER_FOLD)
            ops.gemm_nt(dh, self.wT(p + "mlp.fc1.weight"), dxn2, M, C, hid, hid, hid, C)
            del dh
        # dx += LN2 backward = d(x_mid); the same kernel writes its DropPath-scaled copy, the gradient of the attention branch
        # x_mid = x + s1 * proj(attn(LN1(x)))
        if not ln2_done:
            fuse = bs["s1"] is not None and not _NO_DX2
            dy1 = _empty((M, C), dx.dtype, dev) if fuse else dx
            ops.layernorm_bwd(dxn2, bs["xm"], dx, self.f32(p + "norm2.weight"), bs["m2"], bs["r2"], M, C, C, C, C,
                              dgamma=self.gl(p + "norm2.weight"), dbeta=self.gl(p + "norm2.bias"), **self.lnk(), accumulate=True,
                              dx2=dy1 if fuse else None, dx2_scale=bs["s1"], dx2_rows_per_scale=N, lddx2=C)
        if not fuse:
            dy1 = self._scaled(dx, bs["s1"], N)
        # stages 1-2 (C = 64 / 128, HBM-bound): the weight gradient and the input gradient of a C x C Linear come out of ONE pass over dY
        lin_fuse = dt == torch.bfloat16 and C in (64, 128) and not _NO_LIN_FUSE
        dao = dxn2          # reuse
        if lin_fuse:
            ops.gemm_tn(dy1, bs["ao"], self.g(p + "attn.proj.weight"), M, C, C, C, C, C, colsum=self.g(p + "attn.proj.bias"),
                        dgrad=(self.wT(p + "attn.proj.weight"), dao.view(M, C)))
        else:
            # (stages 3-4: 9-16 output tiles x 32-56 m-splits -- reduced through bf16 partial tiles + a fold instead of atomics: mvlt_gemm_tn_args.partials)
            ops.gemm_tn(dy1, bs["ao"], self.g(p + "attn.proj.weight"), M, C, C, C, C, C, colsum=self.g(p + "attn.proj.bias"), partials=self.S.tn_partials(), defer_fold=_DEFER_FOLD)
            ops.gemm_nt(dy1, self.wT(p + "attn.proj.weight"), dao, M, C, C, C, C, C)
        Mk = bs["Mk"]
        dq = _empty((B, N, C), dt, dev)
        # dK/dV: one query chunk per (batch, head) from B*heads >= 512 on (mvlt_sr_attention_bwd then stores plainly, every
        # element once); only the split case accumulates with atomics and needs the zero fill
        if dt == torch.bfloat16 and ops.sr_attention_bwd_chunks(B, h, N, Mk, dt) == 1:      # (every 256-px stage with B * heads >= 512; round 6: pvlt_medium's stage 3 at batch 64)
            dkv = _empty((B, Mk, 2 * C), dt, dev)              # written once, in the operand dtype
            ops.sr_attention_bwd(bs["q"], bs["kv"], bs["ao"], dao, bs["lse"], dq, dkv, B, h, N, Mk, C, 2 * C, C, 2 * C, 0, C, 64 ** -0.5)
        else:
            dkv32 = pool_zeros((B, Mk, 2 * C), f32, dev)
            ops.sr_attention_bwd(bs["q"], bs["kv"], bs["ao"], dao, bs["lse"], dq, dkv32, B, h, N, Mk, C, 2 * C, C, 2 * C, 0, C, 64 ** -0.5)
            dkv = dkv32.to(dt)
            del dkv32
        # q projection
        dxn1 = dao          # reuse again: d(LN1 output), every row written by the q dgrad
        if lin_fuse:
            ops.gemm_tn(dq, bs["xn1"], self.g(p + "attn.q.weight"), M, C, C, C, C, C, colsum=self.g(p + "attn.q.bias"),
                        dgrad=(self.wT(p + "attn.q.weight"), dxn1.view(M, C)))
        else:
            ops.gemm_tn(dq, bs["xn1"], self.g(p + "attn.q.weight"), M, C, C, C, C, C, colsum=self.g(p + "attn.q.bias"), partials=self.S.tn_partials(), defer_fold=_DEFER_FOLD)
            ops.gemm_nt(dq, self.wT(p + "attn.q.weight"), dxn1, M, C, C, C, C, C)
        gkvw, gkvb = self.g(p + "attn.kv.weight"), self.g(p + "attn.kv.bias")
        wkvT = self.wT(p + "attn.kv.weight")
        if r > 1:
            HWr, pm = bs["HWr"], bs["pm"]
            # text keys come straight from LN1(x)[text rows]
            ops.gemm_tn(dkv, bs["xn1"], gkvw, B * T, 2 * C, C, 2 * C, C, C, a_map=rowmap(T, Mk, HWr), b_map=rowmap(T, N, HW), colsum=gkvb, partials=self.S.tn_partials(), defer_fold=_DEFER_FOLD)
            ops.gemm_nt(dkv, wkvT, dxn1, B * T, C, 2 * C, 2 * C, 2 * C, C, a_map=rowmap(T, Mk, HWr), c_map=rowmap(T, N, HW), R=dxn1)
            # image keys: kv <- LN(sr conv(LN1(x)[image rows]))
            ops.gemm_tn(dkv, bs["kvin"], gkvw, B * HWr, 2 * C, C, 2 * C, C, C, a_map=rowmap(HWr, Mk, 0), colsum=gkvb, partials=self.S.tn_partials(), defer_fold=_DEFER_FOLD)
            dkvin = _empty((B * HWr, C), dt, dev)
            ops.gemm_nt(dkv, wkvT, dkvin, B * HWr, C, 2 * C, 2 * C, 2 * C, C, a_map=rowmap(HWr, Mk, 0))
            dsr = _empty((B * HWr, C), dt, dev)
            ops.layernorm_bwd(dkvin, bs["sr_pre"], dsr, self.f32(p + "attn.norm.weight"), bs["msr"], bs["rsr"], B * HWr, C, C, C, C,
                              dgamma=self.gl(p + "attn.norm.weight"), dbeta=self.gl(p + "attn.norm.bias"), **self.lnk())
            K = r * r * C
            conv_wgrad(self.S, p + "attn.sr.weight", dsr, bs["xn1"], B * HWr, C, K, C, C, pm, r * r, C, colsum=self.g(p + "attn.sr.bias"))
            ops.gemm_nt(dsr, self.wKT(p + "attn.sr.weight"), dxn1, B * HWr, K, C, C, C, C, c_map=pm, R=dxn1)
        else:
            ops.gemm_tn(dkv, bs["xn1"], gkvw, M, 2 * C, C, 2 * C, C, C, colsum=gkvb, partials=self.S.tn_partials(), defer_fold=_DEFER_FOLD)
            ops.gemm_nt(dkv, wkvT, dxn1, M, C, 2 * C, 2 * C, 2 * C, C, R=dxn1)
        # the block below (processed next) scales this gradient by its own MLP-branch DropPath factor first thing when its MLP is not the
        # fused kernel (stages 3-4): norm1's backward writes that scaled copy in the same pass
        prev = self.saved[i]["blocks"][j - 1] if j > 0 else None
        pre = prev is not None and not prev.get("fused_mlp", True) and prev.get("s2") is not None and not _NO_DX2
        self._dy2_pre = _empty((M, C), dx.dtype, dev) if pre else None
        ops.layernorm_bwd(dxn1, bs["x"], dx, self.f32(p + "norm1.weight"), bs["m1"], bs["r1"], M, C, C, C, C,
                          dgamma=self.gl(p + "norm1.weight"), dbeta=self.gl(p + "norm1.bias"), **self.lnk(), accumulate=True,
                          dx2=self._dy2_pre, dx2_scale=prev["s2"] if pre else None, dx2_rows_per_scale=N if pre else 0, lddx2=C if pre else 0)
        bs.clear()
        return dx


class _TrunkFn(torch.autograd.Function):
    """The trunk as ONE autograd node.  Every parameter of the HIP-scheduled part of the model (trunk AND heads) is an
    input of this node: its backward runs after all head nodes, so it can hand autograd the finished slices of the flat
    gradient buffer.  That keeps stock torch machinery working unchanged on top of the side-effect gradient writes:
    AccumulateGrad (and therefore torch DistributedDataParallel's reducer hooks, reference main_vl.py:301), GradScaler,
    clip_grad_norm_ and torch.optim all see ordinary .grad tensors -- which alias the flat buffer, no copies."""

    @staticmethod
    def forward(ctx, model, images, ids, need_grad, *params):
        step = TrunkStep(model, images, ids, need_grad)
        outs = step.forward()
        ctx.step = step
        ctx.pool_token, model._pool_token = model._pool_token, None      # this node's life = the time the step's pooled scratch is owned
        ctx.set_materialize_grads(False)          # unused stage outputs send None, not a zero tensor of their size (138 MB at stage 1)
        ctx.mark_non_differentiable(outs[0])
        return tuple(outs)

    @staticmethod
    def backward(ctx, d1, d2, d3, d4):
        step = ctx.step
        S = step.S
        S.queue_finalize()
        step.backward([None, d2, d3, d4])
        ctx.step = None
        ctx.pool_token = None
        grads = []
        for k, (name, p) in enumerate(S.fn_params):
            gv = S.grad(name)
            if not ctx.needs_input_grad[4 + k]:
                grads.append(None)
            elif p.grad is not None and p.grad.data_ptr() == gv.data_ptr():
                grads.append(None)          # .grad already aliases the flat buffer (accumulation without zero_grad)
            else:
                grads.append(gv)
        return (None, None, None, None, *grads)


# =============================================================================================== heads
class _GradSink:
    """One dense gradient buffer for the last stage's output, shared by its consumers -- MLM / ITM / CLS heads (a few text rows each) and
    the MIM decoder (the image rows).  Every head's backward adds its rows into the same zero-initialised buffer; the first one hands
    it to autograd and the others return None, instead of one zero-filled dense tensor per head plus autograd's dense adds (3 fills and
    2 adds of 50 MB at batch 256).  The trunk node runs after all of them (autograd's dependency count), when the buffer is complete."""

    def __init__(self, store):
        self.S, self.buf = store, None

    def take(self, shape, dtype, dev):
        """-> (buffer, what this caller returns to autograd for the shared input)"""
        if self.buf is None:
            self.buf = self.S.own(torch.zeros(shape, dtype=dtype, device=dev))
            return self.buf, self.buf
        return self.buf, None


def _embed_ln_fwd(model, prefix, A, a_map, rows, lda):
    """head_embed: Linear(512 -> 768) + LN(1e-5) on `rows` rows of A (through a_map)."""
    S, dt, dev = model.store, model.compute_dtype, A.device
    Hd, Cin = model.hidden, model.dims[3]
    pre = _empty((rows, Hd), dt, dev)
    ops.gemm_nt(A, S.comp(prefix + ".0.weight"), pre, rows, Hd, Cin, lda, Cin, Hd, a_map=a_map, bias=S.master(prefix + ".0.bias"))
    y = _empty((rows, Hd), dt, dev)
    mean, rstd = _empty((rows,), torch.float32, dev), _empty((rows,), torch.float32, dev)
    ops.layernorm_fwd(pre, y, S.master(prefix + ".1.weight"), S.master(prefix + ".1.bias"), rows, Hd, Hd, Hd, EPS_DEFAULT, mean=mean, rstd=rstd)
    return y, (pre, mean, rstd)


def _embed_ln_bwd(model, prefix, dy, saved, A, a_map, rows, lda, dA, c_map, accumulate):
    S, dt, dev = model.store, model.compute_dtype, dy.device
    Hd, Cin = model.hidden, model.dims[3]
    pre, mean, rstd = saved
    dpre = _empty((rows, Hd), dt, dev)
    ops.layernorm_bwd(dy, pre, dpre, S.master(prefix + ".1.weight"), mean, rstd, rows, Hd, Hd, Hd, Hd,
                      dgamma=S.grad(prefix + ".1.weight"), dbeta=S.grad(prefix + ".1.bias"))
    ops.gemm_tn(dpre, A, S.grad(prefix + ".0.weight"), rows, Hd, Cin, Hd, lda, Cin, b_map=a_map, colsum=S.grad(prefix + ".0.bias"))
    ops.gemm_nt(dpre, S.extra[prefix + ".0.weight::T"], dA, rows, Cin, Hd, Hd, Hd, Cin, c_map=c_map, R=dA if accumulate else None)


class _ClsHeadFn(torch.autograd.Function):
    """itm / sup_cls / sub_cls: Linear+LN embed of the [CLS] text token, then Linear + extra bias
    (reference libs/pvlt.py:375-388, libs/vl_heads.py:73-104).

    Round 4 ran this B-row chain in fp32 from the fp32 residual stream for one measurement (VERDICT r3 #4 expected the bf16 ITM logits to
    stop depending on rounding luck): the relative error of the logits did not move (tiny224: 1.6e-1 -> 1.8e-1 -- what they show is the
    trunk's ~1e-2 bf16 noise through a 30-fold cancellation, tests/test_model_gpu.py gates them against the dot product's own scale) and the
    exact-f32 MFMA GEMMs on 2 x 6 tiles cost 0.2 ms per step; back in the compute dtype."""

    @staticmethod
    def forward(ctx, x4, model, name, HW, sink):
        S, dt, dev = model.store, model.compute_dtype, x4.device
        B, N, C = x4.shape
        n_out = S.master(name + "_head.linear.weight").shape[0]
        a_map = rowmap(1, N, HW)
        e, saved = _embed_ln_fwd(model, name + "_head_embed", x4, a_map, B, C)
        bias = (S.master(name + "_head.linear.bias") + S.master(name + "_head.linear_bias")).contiguous()
        logits = _empty((B, n_out), torch.float32, dev)
        ops.gemm_nt(e, S.comp(name + "_head.linear.weight"), logits, B, n_out, model.hidden, model.hidden, model.hidden, n_out, bias=bias)
        ctx.pack = (model, name, HW, x4, e, saved, a_map, sink)
        return logits.view(B, 1, n_out)

    @staticmethod
    def backward(ctx, dlogits):
        model, name, HW, x4, e, saved, a_map, sink = ctx.pack
        S, dt, dev = model.store, model.compute_dtype, x4.device
        S.queue_finalize()
        B, N, C = x4.shape
        Hd = model.hidden
        n_out = dlogits.shape[-1]
        n_pad = (n_out + 7) // 8 * 8
        dl = _empty((B, n_pad), dt, dev)
        if n_pad <= 256:
            # padded operand copy + both bias gradients in one launch (six ATen launches before)
            ops.head_grad_prep(dlogits.reshape(B, n_out).float().contiguous(), dl, S.grad(name + "_head.linear.bias"), S.grad(name + "_head.linear_bias"))
        else:
            dl.zero_()
            dl[:, :n_out] = dlogits.reshape(B, n_out).to(dt)
            db = dlogits.reshape(B, n_out).float().sum(0)
            S.grad(name + "_head.linear.bias").add_(db)
            S.grad(name + "_head.linear_bias").add_(db)
        ops.gemm_tn(dl, e, S.grad(name + "_head.linear.weight"), B, n_out, Hd, n_pad, Hd, Hd)
        de = _empty((B, Hd), dt, dev)
        wT = S.extra[name + "_head.linear.weight::T"]          # [768, n_pad]
        ops.gemm_nt(dl, wT, de, B, Hd, n_pad, n_pad, wT.shape[1], Hd)
        dx4, ret = (sink or _GradSink(S)).take(x4.shape, dt, dev)
        _embed_ln_bwd(model, name + "_head_embed", de, saved, x4, a_map, B, C, dx4, a_map, True)
        ctx.pack = None
        S.announce_prefix(name + "_head_embed.", name + "_head.")
        return ret, None, None, None, None


def _mlm_transform_fwd(model, rows_in, R):
    """mlm_head_embed output (R,768) -> BertHeadTransform: dense + erf-GELU + LN(1e-5)."""
    S, dt, dev = model.store, model.compute_dtype, rows_in.device
    Hd = model.hidden
    hp, ga = _empty((R, Hd), dt, dev), _empty((R, Hd), dt, dev)
    ops.gemm_nt(rows_in, S.comp("mlm_head.transform.dense.weight"), ga, R, Hd, Hd, Hd, Hd, Hd,
                bias=S.master("mlm_head.transform.dense.bias"), act=1, H=hp)
    t = _empty((R, Hd), dt, dev)
    mean, rstd = _empty((R,), torch.float32, dev), _empty((R,), torch.float32, dev)
    ops.layernorm_fwd(ga, t, S.master("mlm_head.transform.LayerNorm.weight"), S.master("mlm_head.transform.LayerNorm.bias"),
                      R, Hd, Hd, Hd, EPS_DEFAULT, mean=mean, rstd=rstd)
    return t, (hp, ga, mean, rstd)


def _mlm_transform_bwd(model, dt_, saved, rows_in, R):
    S, dt, dev = model.store, model.compute_dtype, dt_.device
    Hd = model.hidden
    hp, ga, mean, rstd = saved
    dga = _empty((R, Hd), dt, dev)
    ops.layernorm_bwd(dt_, ga, dga, S.master("mlm_head.transform.LayerNorm.weight"), mean, rstd, R, Hd, Hd, Hd, Hd,
                      dgamma=S.grad("mlm_head.transform.LayerNorm.weight"), dbeta=S.grad("mlm_head.transform.LayerNorm.bias"))
    dhp = ops.gelu_bwd(dga, hp, torch.empty_like(dga))             # d(pre-activation) = dga * gelu'(hp)
    ops.gemm_tn(dhp, rows_in, S.grad("mlm_head.transform.dense.weight"), R, Hd, Hd, Hd, Hd, Hd, colsum=S.grad("mlm_head.transform.dense.bias"))
    din = _empty((R, Hd), dt, dev)
    ops.gemm_nt(dhp, S.extra["mlm_head.transform.dense.weight::T"], din, R, Hd, Hd, Hd, Hd, Hd)
    return din


class _MLMFullFn(torch.autograd.Function):
    """Reference-shaped MLM head: logits for every token, (B, T, 30522) (reference libs/pvlt.py:368-370)."""

    @staticmethod
    def forward(ctx, x4, model, HW, sink):
        S, dt, dev = model.store, model.compute_dtype, x4.device
        B, N, C = x4.shape
        T = N - HW
        R = B * T
        a_map = rowmap(T, N, HW)
        e, sv_e = _embed_ln_fwd(model, "mlm_head_embed", x4, a_map, R, C)
        t, sv_t = _mlm_transform_fwd(model, e, R)
        buf = _empty((R, VOCAB_LD), torch.float32, dev)
        ops.gemm_nt(t, S.comp("text_embeddings.word_embeddings.weight"), buf, R, VOCAB, model.hidden, model.hidden, model.hidden, VOCAB_LD,
                    bias=S.master("mlm_head.bias"))
        ctx.pack = (model, HW, x4, e, sv_e, t, sv_t, a_map, sink)
        return buf.view(B, T, VOCAB_LD)[:, :, :VOCAB]

    @staticmethod
    def backward(ctx, dlogits):
        model, HW, x4, e, sv_e, t, sv_t, a_map, sink = ctx.pack
        S, dt, dev = model.store, model.compute_dtype, x4.device
        S.queue_finalize()
        B, N, C = x4.shape
        T = N - HW
        R = B * T
        dl = torch.zeros(R, VOCAB_LD, device=dev, dtype=dt)
        dl[:, :VOCAB] = dlogits.reshape(R, VOCAB).to(dt)
        dx4, ret = (sink or _GradSink(S)).take(x4.shape, dt, dev)
        _mlm_decoder_bwd(model, dl, t, sv_t, e, sv_e, x4, a_map, R, C, dx4, accumulate=True)
        ctx.pack = None
        return ret, None, None, None


def _mlm_decoder_bwd(model, dl, t, sv_t, e, sv_e, A, a_map, R, lda, dA, c_map=None, accumulate=False):
    """shared tail of both MLM paths: dl (R, VOCAB_LD) in the compute dtype -> every MLM-head gradient + dA rows."""
    S, dt, dev = model.store, model.compute_dtype, dl.device
    Hd = model.hidden
    wname = "text_embeddings.word_embeddings.weight"
    # the first writer of the word table's gradient in a pass (bert_embed_bwd adds its lookup rows at the very end): when begin_backward zeroed the whole buffer for THIS
    # pass the 23 M outputs are stored, not added by atomics (mvlt_gemm_tn_args.c_overwrite; a pass that accumulates onto earlier gradients keeps the atomics)
    ops.gemm_tn(dl, t, S.grad(wname), R, VOCAB, Hd, VOCAB_LD, Hd, Hd, colsum=S.grad("mlm_head.bias"),
                overwrite=dt == torch.bfloat16 and S.all_zeroed_this_pass and wname not in S.touched_this_pass and not _NO_OVERWRITE)
    S.touched_this_pass.add(wname)
    wT = S.extra[wname + "::T"]                                   # [768, VOCAB_LD], zero padded
    if dt == torch.bfloat16:
        # few output tiles (R x 768), K = 30528: cut K over 4 workgroups per tile, partial sums meet in an fp32 buffer
        dtr32 = pool_zeros((R, Hd), torch.float32, dev)
        ops.gemm_nt(dl, wT, dtr32, R, Hd, VOCAB_LD, VOCAB_LD, VOCAB_LD, Hd, split_k=4)
        dtr = dtr32.to(dt)
    else:
        dtr = _empty((R, Hd), dt, dev)
        ops.gemm_nt(dl, wT, dtr, R, Hd, VOCAB_LD, VOCAB_LD, VOCAB_LD, Hd)
    de = _mlm_transform_bwd(model, dtr, sv_t, e, R)
    _embed_ln_bwd(model, "mlm_head_embed", de, sv_e, A, a_map, R, lda, dA, c_map if c_map is not None else a_map, accumulate)
    # final now: the head's own parameters.  The tied decoder weight is the word-embedding table, which bert_embed_bwd still adds
    # to at the very end of the pass: it travels with the leftovers.
    S.announce_prefix("mlm_head_embed.", "mlm_head.")


class _MLMFusedFn(torch.autograd.Function):
    """MLM head + CrossEntropyLoss(ignore_index=-1) on the selected rows only.  `positions` (int32, ascending flat
    b*T+t indices with label != -1) is the masked-index selection; rows CE would ignore are never computed."""

    @staticmethod
    def forward(ctx, x4, model, HW, positions, labels_sel, sink):
        S, dt, dev = model.store, model.compute_dtype, x4.device
        B, N, C = x4.shape
        T = N - HW
        R = positions.numel()
        rows = _empty((R, C), dt, dev)
        tmap = rowmap(T, N, HW)
        ops.gather_rows(x4, positions, rows, R, C, C, src_map=tmap)
        e, sv_e = _embed_ln_fwd(model, "mlm_head_embed", rows, None, R, C)
        t, sv_t = _mlm_transform_fwd(model, e, R)
        logits = _empty((R, VOCAB_LD), torch.float32, dev)
        ops.gemm_nt(t, S.comp("text_embeddings.word_embeddings.weight"), logits, R, VOCAB, model.hidden, model.hidden, model.hidden, VOCAB_LD,
                    bias=S.master("mlm_head.bias"))
        lse = _empty((R,), torch.float32, dev)
        acc = pool_zeros((2,), torch.float32, dev)                     # [loss_sum, count]; pooled scratch lives until the next forward
        ops.cross_entropy_fwd(logits, labels_sel, lse, acc[0:1], acc[1:2], R, VOCAB, VOCAB_LD)
        ctx.pack = (model, HW, x4.shape, rows, e, sv_e, t, sv_t, logits, lse, acc, positions, labels_sel, tmap, sink)
        return acc[0] / acc[1]          # mean over selected rows (NaN when none, like torch)

    @staticmethod
    def backward(ctx, gloss):
        model, HW, xshape, rows, e, sv_e, t, sv_t, logits, lse, acc, positions, labels_sel, tmap, sink = ctx.pack
        S, dt, dev = model.store, model.compute_dtype, rows.device
        S.queue_finalize()
        B, N, C = xshape
        R = positions.numel()
        dl = _empty((R, VOCAB_LD), dt, dev)
        gs = gloss.reshape(1).float().contiguous()
        ops.cross_entropy_bwd(logits, labels_sel, lse, gs, acc[1:2], dl, R, VOCAB, VOCAB_LD, VOCAB_LD)
        drows = _empty((R, C), dt, dev)
        _mlm_decoder_bwd(model, dl, t, sv_t, e, sv_e, rows, None, R, C, drows, c_map=None)
        dx4, ret = (sink or _GradSink(S)).take(xshape, dt, dev)
        ops.scatter_rows(drows, positions, dx4, R, C, C, dst_map=tmap, accumulate=True)
        ctx.pack = None
        return ret, None, None, None, None, None


# =============================================================================================== top level
class _HostCount:
    """A device int32 counter on its way to the host without draining the launch queue: an asynchronous copy into pinned
    memory plus an event recorded right behind it.  `get()` waits for that event only -- the kernels queued after it (the
    whole trunk forward, when the selection is the first launch of the step) keep the GPU busy meanwhile."""
    _pins = {}

    def __init__(self, cnt):
        key = (cnt.device.index, cnt.numel())
        pin = self._pins.get(key)
        if pin is None:
            pin = self._pins[key] = torch.empty(cnt.numel(), dtype=cnt.dtype).pin_memory()
        self.pin = pin
        pin.copy_(cnt, non_blocking=True)
        self.ev = torch.cuda.Event()
        self.ev.record()

    def get(self):
        self.ev.synchronize()
        return int(self.pin[0])


def run_forward(model, images, ids, mlm_labels=None, mlm_positions=None, mlm_count=None, t2i_target=None):
    """mlm_labels: (B, T) int64 with -1 = not selected -> fused masked-row MLM head + loss (`mlm_loss`).
    t2i_target: (B, 3, S, S) fp32 clean image -> the MIM decoder returns its SmoothL1 loss (`t2i_loss`) instead of `t2i_logits`
    (training: the image-sized prediction is never materialised, engine_grid_masking.py:99 is fused behind vl_heads.py:163-165).
    mlm_positions: optional precomputed selection (ascending flat indices, int32, on the device).
    mlm_count: optional number of selected positions as a host int (the engine counts on the host when the loader hands it
    CPU labels; the device prefetcher brings it along) -- without it the count comes back through `_HostCount`."""
    S = model.store
    dev = images.device
    grad_on = _begin_pass(model, dev)
    lt = model.loss_type
    sel = None
    if lt['mlm'] and mlm_labels is not None and mlm_positions is None:
        # masked-index selection (bit-exact vs torch.nonzero): first launch of the step, so that its count is on the host long
        # before the MLM head needs it to size its launches
        flat = mlm_labels.reshape(-1).contiguous()
        idx = torch.empty(flat.numel(), device=dev, dtype=torch.int32)
        cnt = pool_zeros((1,), torch.int32, dev)
        ops.masked_select(flat, idx, cnt)
        sel = (idx, mlm_count if mlm_count is not None else _HostCount(cnt))
    S.refresh(model._transposed, model._conv_perm, model._conv3)
    x1, x2, x3, x4 = _TrunkFn.apply(model, images, ids, grad_on, *[p for _, p in S.fn_params])
    sink = _GradSink(S) if grad_on else None            # the heads' common gradient buffer for x4
    B = images.shape[0]
    side4 = images.shape[2] // model.patch_size // 8
    HW4 = side4 * side4
    out = dict(mlm_logits=None, itm_logits=None, sup_cls_logits=None, sub_cls_logits=None, t2i_logits=None)
    if lt['mlm']:
        if mlm_labels is not None:
            flat = mlm_labels.reshape(-1).contiguous()
            if sel is not None:
                idx, n = sel
                mlm_positions = idx[: (n if isinstance(n, int) else n.get())]
            if mlm_positions.numel() == 0:
                # nothing selected in this batch: CrossEntropyLoss(ignore_index=-1) averages over zero rows -> NaN, and no row
                # sends a gradient back (reference engine_grid_masking.py:84 behaves the same way)
                out["mlm_loss"] = torch.full((), float("nan"), device=dev)
            else:
                labels_sel = flat[mlm_positions.long()].contiguous()
                out["mlm_loss"] = _MLMFusedFn.apply(x4, model, HW4, mlm_positions.contiguous(), labels_sel, sink)
            out["mlm_positions"] = mlm_positions
        else:
            out["mlm_logits"] = _MLMFullFn.apply(x4, model, HW4, sink)
    if lt['itm']:
        out["itm_logits"] = _ClsHeadFn.apply(x4, model, "itm", HW4, sink)
    if lt['cls']:
        out["sup_cls_logits"] = _ClsHeadFn.apply(x4, model, "sup_cls", HW4, sink)
        out["sub_cls_logits"] = _ClsHeadFn.apply(x4, model, "sub_cls", HW4, sink)
    if lt['t2i']:
        from .mim import mim_head
        if grad_on and not model.training:
            raise NotImplementedError("MIM decoder backward with eval-mode BatchNorm is not scheduled (no reference config needs it)")
        sides = tuple(images.shape[2] // model.patch_size // (2 ** i) for i in (1, 2, 3))
        fuse_loss = (t2i_target is not None and t2i_target.dtype == torch.float32 and t2i_target.shape == (B, 3, 8 * sides[0], 8 * sides[0])
                     and ops.upsample_l1_ok(sides[0], 8) and not _NO_T2I_FUSE)
        if fuse_loss:
            out["t2i_loss"] = mim_head(model, x2, x3, x4, sides, grad_on, sink, t2i_target.contiguous())
        else:
            out["t2i_logits"] = mim_head(model, x2, x3, x4, sides, grad_on, sink)
    return out


def _begin_pass(model, dev):
    """common head of every forward entry: flat store on the device, one fill for all of this step's zero-initialised scratch"""
    S = model.store
    S.ensure(dev)
    grad_on = torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters())
    model._pool_token = ZeroPool.of(dev).reset(grad_on)
    if grad_on:
        S.new_pass()                                              # a backward that raised must not poison this pass (FlatStore.new_pass)
    return grad_on


def run_trunk(model, images, ids):
    """The trunk alone, behind `forward_pyramid_features_vl` (reference libs/pvlt.py:322-356): the four stage outputs as
    (B, HW_i + T, C_i) token buffers, image tokens first (differentiable: one autograd node, like in `run_forward`)."""
    S = model.store
    grad_on = _begin_pass(model, images.device)
    S.refresh(model._transposed, model._conv_perm, model._conv3)
    return _TrunkFn.apply(model, images, ids, grad_on, *[p for _, p in S.fn_params])
