"""Thin Python calls over the C ABI (include/mvlt_hip.h).  Every function launches HIP kernels asynchronously on
torch's current stream; tensors are only used as device buffers.  No fallbacks: a missing library or a CPU tensor
raises (mvlt_amd/_lib.py)."""
import ctypes as C

import torch

from . import _lib as L
from ._lib import DT, check, patchmap, ptr, rowmap, stream_ptr

_ID = rowmap()


def gemm_nt(A, B, C_out, M, N, K, lda, ldb, ldc, *, a_map=None, c_map=None, bias=None, act=0, H=None,
            row_scale=None, rows_per_scale=0, R=None, col_sum=None, col_sumsq=None, col_copies=0, split_k=0, post_ln=None):
    """C[M,N] = epi(A[M,K] @ B[N,K]^T); see mvlt_gemm_nt in include/mvlt_hip.h.  post_ln = (gamma, beta, eps, y, mean, rstd): LayerNorm of the
    finished output rows rides on the epilogue (N == 64 / 128, bf16 operands, R given)."""
    assert A.dtype == B.dtype and A.dtype in DT and (C_out.dtype in DT or (C_out.dtype == torch.float16 and col_sum is not None))
    if bias is not None:
        assert bias.dtype == torch.float32
    if row_scale is not None:
        assert row_scale.dtype == torch.float32
    if R is not None:
        assert R.dtype == C_out.dtype or (R.dtype == torch.float32 and C_out.dtype == torch.bfloat16)       # (the second: mvlt_gemm_nt_args.r_fp32)
    if H is not None:
        assert H.dtype == C_out.dtype
    if col_sum is not None:
        assert col_sum.dtype == torch.float32 and col_sumsq is not None and col_sumsq.dtype == torch.float32
    a = L.GemmNTArgs(ptr(A), ptr(B), ptr(C_out), M, N, K, lda, ldb, ldc, DT[A.dtype], 2 if C_out.dtype == torch.float16 else DT[C_out.dtype],
                     a_map or _ID, c_map or _ID, ptr(bias), act, ptr(H), ptr(row_scale), rows_per_scale, ptr(R),
                     ptr(col_sum), ptr(col_sumsq), col_copies, split_k)
    if R is not None and R.dtype != C_out.dtype:
        a.r_fp32 = 1
    if post_ln is not None:
        g, b, eps, y, mean, rstd = post_ln
        assert g.dtype == b.dtype == mean.dtype == rstd.dtype == torch.float32 and y.dtype == torch.bfloat16 and y.is_contiguous()
        a.post_y, a.post_ld, a.post_gamma, a.post_beta, a.post_eps, a.post_mean, a.post_rstd = ptr(y), y.shape[-1], ptr(g), ptr(b), eps, ptr(mean), ptr(rstd)
    check(L.lib.mvlt_gemm_nt(C.byref(a), stream_ptr()), "mvlt_gemm_nt")
    return C_out


def tn_fold_flush(partials=None):
    """fold the partial-tile reductions that gemm_tn(partials=..., defer_fold=True) left pending in that scratch (None: in every scratch): one launch per scratch on the
    producers' stream, which the current stream waits for if it is another one; none pending: none"""
    check(L.lib.mvlt_tn_fold_flush(ptr(partials), stream_ptr()), "mvlt_tn_fold_flush")


def tn_fold_discard(partials=None):
    """drop the pending deferred folds of that scratch (None: all) without running them (the start of a backward pass: whatever its store has pending then belongs to a
    pass that was abandoned)"""
    check(L.lib.mvlt_tn_fold_discard(ptr(partials)), "mvlt_tn_fold_discard")


def weight_prep(desc_dev, blk_dev, ndesc, total_blocks, dtype, blk_desc=None):
    """desc_dev: uint8 device tensor holding ndesc packed mvlt_prep_desc; blk_dev: int32 device tensor [ndesc + 1]."""
    check(L.lib.mvlt_weight_prep(C.c_void_p(desc_dev.data_ptr()), C.c_void_p(blk_dev.data_ptr()), ndesc, total_blocks,
                                 C.c_void_p(blk_desc.data_ptr() if blk_desc is not None else None), DT[dtype], stream_ptr()),
          "mvlt_weight_prep")


def gemm_tn(A, B, C_out, M, N1, N2, lda, ldb, ldc, *, a_map=None, b_map=None, colsum=None, splits=0, taps=0, seg=0, dgrad=None, partials=None, defer_fold=False, overwrite=False):
    """C[N1,N2] += A[M,N1]^T @ B[M,N2] (fp32 atomics); colsum[N1] += A.sum(0).  taps > 1: logical column tap*seg + c is
    accumulated at column c*taps + tap (conv weight gradients straight into the [out][cin][kh][kw] layout).
    dgrad = (W^T [N2][N1] bf16, out [M, N2] bf16): the Linear's input gradient out = A @ W from the same pass over A (N1 == N2 in {64, 128}).
    partials = a scratch tensor (the largest single launch of the model needs 37 MiB; deferring launches share it region by region: FlatStore holds 256 MiB): split reductions leave as bf16 partial tiles + an ordered fold instead of fp32 atomics where the
    library has that mode (mvlt_gemm_tn_args.partials in include/mvlt_hip.h: whole 256 x 256 tiles, >= 8 m-splits on the 128-wide kernel, conv3x3 weight gradients); deterministic."""
    assert A.dtype == B.dtype and A.dtype in DT and C_out.dtype == torch.float32
    if colsum is not None:
        assert colsum.dtype == torch.float32
    a_map, b_map = a_map or _ID, b_map or _ID
    if dgrad is not None:
        wt, dx = dgrad
        assert A.dtype == torch.bfloat16 and wt.dtype == dx.dtype == torch.bfloat16 and N1 == N2 and N1 in (64, 128) and taps <= 1
        assert wt.is_contiguous() and tuple(wt.shape) == (N2, N1) and dx.stride(-1) == 1
        # dx may be a row-strided view (a column slice of a wider buffer): its own row pitch goes to the kernel (N2 was passed);
        # the fused kernel walks plain rows on both operands
        dgrad_ld = dx.stride(-2) if dx.dim() >= 2 else N2
        assert dgrad_ld % 8 == 0 and dgrad_ld >= N2, dgrad_ld
        assert a_map.mode == 0 and a_map.rows_per_batch == 0 and b_map.mode == 0 and b_map.rows_per_batch == 0, "gemm_tn(dgrad=...): plain row maps only"
        a = L.GemmTNArgs(ptr(A), ptr(B), ptr(C_out), M, N1, N2, lda, ldb, ldc, DT[A.dtype], a_map, b_map, ptr(colsum), splits, None, 0, 0, 0,
                         ptr(wt), ptr(dx), dgrad_ld)
        check(L.lib.mvlt_gemm_tn(C.byref(a), stream_ptr()), "mvlt_gemm_tn")
        return C_out
    if N1 <= 64 < N2 and b_map.mode == 0 and taps <= 1 and partials is None:          # (with a partial-tile scratch the 64 x 128 tile of the LDS-DMA kernel takes the shape as it is)
        # the kernel's tile is 128 (N1 side) x 64/128 (N2 side): give the narrow operand the 64-wide side by computing
        # C^T = B^T A and storing it transposed; the bias gradient becomes the column sum of the (now) B operand
        a = L.GemmTNArgs(ptr(B), ptr(A), ptr(C_out), M, N2, N1, ldb, lda, ldc, DT[A.dtype], b_map, a_map, None, splits, ptr(colsum), 1, 0, 0)
    else:
        a = L.GemmTNArgs(ptr(A), ptr(B), ptr(C_out), M, N1, N2, lda, ldb, ldc, DT[A.dtype], a_map, b_map, ptr(colsum), splits, None, 0, taps, seg)
        if partials is not None:       # scratch for the atomic-free reduction of whole 256 x 256 output tiles (mvlt_gemm_tn_args.partials); ignored for other shapes
            a.partials, a.partials_bytes = ptr(partials), partials.numel() * partials.element_size()
            a.defer_fold = 1 if defer_fold else 0       # the caller promises tn_fold_flush() before anything reads C_out
    if overwrite:
        assert not a.trans_c and taps <= 1 and N2 % 4 == 0 and ldc % 4 == 0 and A.dtype == torch.bfloat16, "gemm_tn(overwrite=True): bf16, plain output layout, N2 and ldc multiples of 4"
        a.c_overwrite = 1                               # C_out holds zeros (the caller's word): one m-split, plain stores instead of atomics
    check(L.lib.mvlt_gemm_tn(C.byref(a), stream_ptr()), "mvlt_gemm_tn")
    return C_out


LN_CHAIN_WIDTHS = (64, 128, 320, 512, 768)     # widths with a fixed-geometry forward kernel: these can chain a second LayerNorm


def layernorm_fwd(x, y, gamma, beta, rows, Cdim, ldx, ldy, eps, *, mean=None, rstd=None, add=None, add_rows=0,
                  x_map=None, y_map=None, chain=None):
    """chain = (gamma2, beta2, eps2, y2, mean2, rstd2): y2 (bf16, rows laid out like y) = LayerNorm(y) with the second parameter set,
    from the same pass; mean2 / rstd2 are indexed by y's physical row.  Only for Cdim in LN_CHAIN_WIDTHS."""
    assert x.dtype in DT and y.dtype in DT and gamma.dtype == torch.float32 and beta.dtype == torch.float32
    tail = (None, None, None, 0.0, None, None)
    if chain is not None:
        g2, b2, eps2, y2, m2, r2 = chain
        assert Cdim in LN_CHAIN_WIDTHS and y2.dtype == torch.bfloat16 and g2.dtype == b2.dtype == m2.dtype == r2.dtype == torch.float32
        tail = (ptr(y2), ptr(g2), ptr(b2), eps2, ptr(m2), ptr(r2))
    a = L.LayerNormArgs(ptr(x), ptr(y), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(add), add_rows,
                        rows, Cdim, ldx, ldy, x_map or _ID, y_map or _ID, eps, DT[x.dtype], DT[y.dtype], *tail)
    check(L.lib.mvlt_layernorm_fwd(C.byref(a), stream_ptr()), "mvlt_layernorm_fwd")
    return y


def layernorm_bwd(dy, x, dx, gamma, mean, rstd, rows, Cdim, lddy, ldx, lddx, *, dgamma=None, dbeta=None,
                  dy_map=None, x_map=None, dx_map=None, accumulate=False, dx2=None, dx2_scale=None, dx2_rows_per_scale=0, lddx2=0,
                  copies=1, copy_stride=0):
    assert dy.dtype in DT and x.dtype in DT and dx.dtype in DT
    assert dx2 is None or (dx2.dtype == dy.dtype and dx2_scale is not None and dx2_scale.dtype == torch.float32 and dx2_rows_per_scale > 0)
    a = L.LayerNormBwdArgs(ptr(dy), ptr(x), ptr(dx), ptr(gamma), ptr(mean), ptr(rstd), ptr(dgamma), ptr(dbeta),
                           rows, Cdim, lddy, ldx, lddx, dy_map or _ID, x_map or _ID, dx_map or _ID,
                           1 if accumulate else 0, DT[dy.dtype], DT[x.dtype], DT[dx.dtype],
                           ptr(dx2), ptr(dx2_scale), dx2_rows_per_scale, lddx2, copies, copy_stride)
    check(L.lib.mvlt_layernorm_bwd(C.byref(a), stream_ptr()), "mvlt_layernorm_bwd")
    return dx


L.lib.mvlt_fold_copies.argtypes = [C.c_void_p, C.c_int, C.c_long, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]


def fold_copies(arena, copies, stride, dst_index, j0, j1, dst):
    assert arena.dtype == torch.float32 and dst.dtype == torch.float32 and dst_index.dtype == torch.int32
    check(L.lib.mvlt_fold_copies(ptr(arena), copies, stride, ptr(dst_index), j0, j1, ptr(dst), stream_ptr()), "mvlt_fold_copies")


L.lib.mvlt_batch_sum.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]


def batch_sum(x, out, B, R, Cdim, batch_stride_rows, ld, acc2=None, split=0):
    """out[r] = sum over the batch of x[b, r]; with acc2, rows r >= split are added to acc2[r - split] instead"""
    assert out.dtype == torch.float32 and (acc2 is None or (acc2.dtype == torch.float32 and acc2.is_contiguous()))
    check(L.lib.mvlt_batch_sum(ptr(x), ptr(out), B, R, Cdim, batch_stride_rows, ld, DT[x.dtype], ptr(acc2), split, stream_ptr()), "mvlt_batch_sum")
    return out


def sr_attention_fwd(Q, KV, O, lse, B, H, N, M, ldq, ldkv, ldo, k_off, v_off, scale):
    assert Q.dtype == KV.dtype == O.dtype and Q.dtype in DT
    a = L.AttnArgs(ptr(Q), ptr(KV), ptr(O), ptr(lse), B, H, N, M, ldq, ldkv, ldo, k_off, v_off, scale, DT[Q.dtype])
    check(L.lib.mvlt_sr_attention_fwd(C.byref(a), stream_ptr()), "mvlt_sr_attention_fwd")
    return O


def sr_attention_bwd_chunks(B, H, N, M, dtype):
    """query chunks per (batch, head) the backward would use with an fp32 dKV; 1: hand it a bf16 dKV instead (every element stored once: no zero fill, no cast)"""
    return int(L.lib.mvlt_sr_attention_bwd_chunks(B, H, N, M, DT[dtype]))


def sr_attention_bwd(Q, KV, O, dO, lse, dQ, dKV, B, H, N, M, ldq, ldkv, ldo, lddkv, k_off, v_off, scale):
    assert dKV.dtype == torch.float32 or (dKV.dtype == torch.bfloat16 and Q.dtype == torch.bfloat16)
    a = L.AttnBwdArgs(ptr(Q), ptr(KV), ptr(O), ptr(dO), ptr(lse), ptr(dQ), ptr(dKV), B, H, N, M,
                      ldq, ldkv, ldo, lddkv, k_off, v_off, scale, DT[Q.dtype], DT[dKV.dtype])
    check(L.lib.mvlt_sr_attention_bwd(C.byref(a), stream_ptr()), "mvlt_sr_attention_bwd")
    return dQ, dKV


# ------------------------------------------------------------------ helpers of csrc/elementwise.hip
_vp, _i, _l, _f = C.c_void_p, C.c_int, C.c_long, C.c_float
L.lib.mvlt_bert_embed_fwd.argtypes = [_vp] * 7 + [_f, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp]
L.lib.mvlt_bert_embed_bwd.argtypes = [_vp] * 7 + [_f] + [_vp] * 7 + [_i, _i, _i, _i, _vp]
L.lib.mvlt_patchify.argtypes = [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]
L.lib.mvlt_masked_select.argtypes = [_vp, _i, _l, _vp, _vp, _vp]
L.lib.mvlt_gather_rows.argtypes = [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp]
L.lib.mvlt_scatter_rows.argtypes = [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _vp]
L.lib.mvlt_cross_entropy_fwd.argtypes = [_vp, _vp, _l, _vp, _vp, _vp, _i, _i, _i, _i, _vp]
L.lib.mvlt_cross_entropy_bwd.argtypes = [_vp, _vp, _l, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]
L.lib.mvlt_adamw_step.argtypes = [_vp, _vp, _vp, _vp, _vp, _l, _vp, _vp, _vp]
L.lib.mvlt_cast_bf16.argtypes = [_vp, _vp, _l, _vp]
L.lib.mvlt_transpose_cast.argtypes = [_vp, _vp, _i, _i, _i, _i, _vp]


def _p(t):
    return None if t is None else t.data_ptr()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.MVLTError("mvlt_amd ops need CUDA/HIP tensors (no CPU fallback)")


def bert_embed_fwd(ids, word, pos, type0, gamma, beta, keep, drop_p, y, mean, rstd, rows, T, eps):
    _need_cuda(ids, word, y)
    assert ids.dtype == torch.int64 and (keep is None or keep.dtype == torch.uint8)
    check(L.lib.mvlt_bert_embed_fwd(_p(ids), _p(word), _p(pos), _p(type0), _p(gamma), _p(beta), _p(keep), drop_p,
                                    _p(y), _p(mean), _p(rstd), rows, T, word.shape[1], eps, DT[y.dtype], stream_ptr()),
          "mvlt_bert_embed_fwd")
    return y


def bert_embed_bwd(dy, ids, word, pos, type0, gamma, keep, drop_p, mean, rstd, dword, dpos, dtype0, dgamma, dbeta, rows, T):
    _need_cuda(dy, ids, dword)
    check(L.lib.mvlt_bert_embed_bwd(_p(dy), _p(ids), _p(word), _p(pos), _p(type0), _p(gamma), _p(keep), drop_p,
                                    _p(mean), _p(rstd), _p(dword), _p(dpos), _p(dtype0), _p(dgamma), _p(dbeta),
                                    rows, T, word.shape[1], DT[dy.dtype], stream_ptr()), "mvlt_bert_embed_bwd")


L.lib.mvlt_head_grad_prep.argtypes = [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp]


def head_grad_prep(dlogits, dl, db1, db2=None):
    """dl[B, n_pad] = dlogits[B, n] (zero padded, dl's dtype); db1 (and db2) += dlogits.sum(0)"""
    assert dlogits.dtype == torch.float32 and dlogits.is_contiguous() and dl.is_contiguous() and dl.dtype in DT and db1.dtype == torch.float32
    B, n = dlogits.shape
    check(L.lib.mvlt_head_grad_prep(_p(dlogits), B, n, dl.shape[1], _p(dl), _p(db1), _p(db2), DT[dl.dtype], stream_ptr()), "mvlt_head_grad_prep")


def patchify(img, out, B, Cin, H, W, k):
    _need_cuda(img, out)
    assert img.dtype == torch.float32 and img.is_contiguous()
    check(L.lib.mvlt_patchify(_p(img), _p(out), B, Cin, H, W, k, DT[out.dtype], stream_ptr()), "mvlt_patchify")
    return out


def masked_select(labels, idx, count, ignore_index=-1):
    _need_cuda(labels, idx, count)
    assert labels.dtype == torch.int64 and idx.dtype == torch.int32 and count.dtype == torch.int32 and labels.is_contiguous()
    check(L.lib.mvlt_masked_select(_p(labels), labels.numel(), ignore_index, _p(idx), _p(count), stream_ptr()), "mvlt_masked_select")


def gather_rows(src, idx, dst, rows, Cdim, ld_src, src_map=None):
    _need_cuda(src, idx, dst)
    assert idx.dtype == torch.int32 and src.dtype == dst.dtype
    m = C.byref(src_map) if src_map is not None else None
    check(L.lib.mvlt_gather_rows(_p(src), _p(idx), _p(dst), rows, Cdim, ld_src, m, DT[src.dtype], stream_ptr()), "mvlt_gather_rows")
    return dst


def scatter_rows(src, idx, dst, rows, Cdim, ld_dst, dst_map=None, accumulate=False):
    _need_cuda(src, idx, dst)
    assert idx.dtype == torch.int32 and src.dtype == dst.dtype
    m = C.byref(dst_map) if dst_map is not None else None
    check(L.lib.mvlt_scatter_rows(_p(src), _p(idx), _p(dst), rows, Cdim, ld_dst, m, 1 if accumulate else 0, DT[src.dtype], stream_ptr()),
          "mvlt_scatter_rows")
    return dst


def cross_entropy_fwd(logits, labels, lse, loss_sum, count, rows, V, ld, ignore_index=-1):
    _need_cuda(logits, labels, lse)
    assert labels.dtype == torch.int64
    check(L.lib.mvlt_cross_entropy_fwd(_p(logits), _p(labels), ignore_index, _p(lse), _p(loss_sum), _p(count), rows, V, ld,
                                       DT[logits.dtype], stream_ptr()), "mvlt_cross_entropy_fwd")


def cross_entropy_bwd(logits, labels, lse, gscale, count, dlogits, rows, V, ld, ldd, ignore_index=-1):
    _need_cuda(logits, labels, dlogits)
    check(L.lib.mvlt_cross_entropy_bwd(_p(logits), _p(labels), ignore_index, _p(lse), _p(gscale), _p(count), _p(dlogits), rows, V, ld, ldd,
                                       DT[logits.dtype], DT[dlogits.dtype], stream_ptr()), "mvlt_cross_entropy_bwd")


def adamw_step(p, g, m, v, p16, n, hp, decay_mask=None):
    _need_cuda(p, g, m, v, hp)
    assert decay_mask is None or decay_mask.dtype == torch.uint8
    check(L.lib.mvlt_adamw_step(_p(p), _p(g), _p(m), _p(v), _p(p16), n, _p(hp), _p(decay_mask), stream_ptr()), "mvlt_adamw_step")


L.lib.mvlt_smooth_l1_fwd.argtypes = [_vp, _vp, _l, _vp, _vp]
L.lib.mvlt_smooth_l1_bwd.argtypes = [_vp, _vp, _l, _vp, _vp, _vp]


def smooth_l1_fwd(pred, target, loss_sum):
    _need_cuda(pred, target, loss_sum)
    assert pred.dtype == torch.float32 and target.dtype == torch.float32 and pred.is_contiguous() and target.is_contiguous()
    check(L.lib.mvlt_smooth_l1_fwd(_p(pred), _p(target), pred.numel(), _p(loss_sum), stream_ptr()), "mvlt_smooth_l1_fwd")


def smooth_l1_bwd(pred, target, gscale, grad):
    _need_cuda(pred, target, gscale, grad)
    check(L.lib.mvlt_smooth_l1_bwd(_p(pred), _p(target), pred.numel(), _p(gscale), _p(grad), stream_ptr()), "mvlt_smooth_l1_bwd")


def cast_bf16(src, dst, n):
    _need_cuda(src, dst)
    check(L.lib.mvlt_cast_bf16(_p(src), _p(dst), n, stream_ptr()), "mvlt_cast_bf16")


L.lib.mvlt_row_scale.argtypes = [_vp, _vp, _i, _l, _i, _vp, _i, _vp]


def row_scale(x, scale, rows_per_scale, M, Cdim, out):
    _need_cuda(x, scale, out)
    assert x.is_contiguous() and out.is_contiguous() and scale.dtype == torch.float32 and x.dtype == out.dtype
    check(L.lib.mvlt_row_scale(_p(x), _p(scale), rows_per_scale, M, Cdim, _p(out), DT[x.dtype], stream_ptr()), "mvlt_row_scale")


def transpose_cast(w, out, R, Ccols, ld_out):
    _need_cuda(w, out)
    assert w.dtype == torch.float32 and w.is_contiguous()
    check(L.lib.mvlt_transpose_cast(_p(w), _p(out), R, Ccols, ld_out, DT[out.dtype], stream_ptr()), "mvlt_transpose_cast")


# ------------------------------------------------------------------ MIM decoder helpers (csrc/mim.hip)
L.lib.mvlt_col_stats.argtypes = [_vp, _i, _l, _i, _vp, _vp, _vp]
L.lib.mvlt_bn_finalize.argtypes = [_vp, _vp, _i, _l, _i, _f, _f, _vp, _vp, _vp, _vp, _vp]
L.lib.mvlt_bn_norm.argtypes = [_vp, _i, _i, _vp, _vp, _vp, _vp, _l, _i, _vp, _i, _i, _vp, _i, _i, _vp]
L.lib.mvlt_bn_bwd_reduce.argtypes = [_vp, _i, _vp, _i, _i, _vp, _vp, _l, _i, _vp, _vp, _i, _vp]
L.lib.mvlt_bn_bwd_apply.argtypes = [_vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _l, _i, _vp, _i, _vp, _vp, _i, _i, _vp]
ZDT = {torch.float32: 1, torch.float16: 2}       # the pre-BatchNorm conv output z: fp32, or fp16 on the bf16 path (mvlt_gemm_nt out_dtype 2)
L.lib.mvlt_ew_mul.argtypes = [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _l, _i, _i, _vp, _i, _i, _vp]
L.lib.mvlt_upsample_fwd.argtypes = [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _vp]
L.lib.mvlt_upsample_bwd.argtypes = [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp]


def col_stats(z, ldz, M, Cdim, s, ss):
    _need_cuda(z, s, ss)
    check(L.lib.mvlt_col_stats(_p(z), ldz, M, Cdim, _p(s), _p(ss), stream_ptr()), "mvlt_col_stats")


def bn_finalize(s, ss, M, Cdim, eps, momentum, mean, rstd, running_mean=None, running_var=None, copies=1):
    check(L.lib.mvlt_bn_finalize(_p(s), _p(ss), copies, M, Cdim, eps, momentum, _p(mean), _p(rstd), _p(running_mean), _p(running_var), stream_ptr()),
          "mvlt_bn_finalize")


def bn_norm(z, ldz, mean, rstd, gamma, beta, M, Cdim, y32=None, ld32=0, y16=None, ld16=0):
    check(L.lib.mvlt_bn_norm(_p(z), ldz, ZDT[z.dtype], _p(mean), _p(rstd), _p(gamma), _p(beta), M, Cdim, _p(y32), ld32, ZDT[y32.dtype] if y32 is not None else 1, _p(y16), ld16,
                             DT[y16.dtype] if y16 is not None else 0, stream_ptr()), "mvlt_bn_norm")


L.lib.mvlt_bn_finalize_norm.argtypes = [_vp, _i, _vp, _vp, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _vp, _i, _i, _vp, _i, _vp]


def bn_finalize_norm(z, ldz, s, ss, copies, eps, momentum, mean, rstd, running_mean, running_var, gamma, beta, M, Cdim, y32=None, ld32=0, y16=None, ld16=0):
    assert z.dtype == torch.float16 and (y16 is None or y16.dtype == torch.bfloat16)
    check(L.lib.mvlt_bn_finalize_norm(_p(z), ldz, _p(s), _p(ss), copies, eps, momentum, _p(mean), _p(rstd), _p(running_mean), _p(running_var), _p(gamma), _p(beta), M, Cdim,
                                      _p(y32), ld32, ZDT[y32.dtype] if y32 is not None else 1, _p(y16), ld16, stream_ptr()), "mvlt_bn_finalize_norm")


def bn_bwd_reduce(dy, lddy, z, ldz, mean, rstd, M, Cdim, s1, s2):
    assert dy.dtype in DT and z.dtype in ZDT
    check(L.lib.mvlt_bn_bwd_reduce(_p(dy), lddy, _p(z), ldz, ZDT[z.dtype], _p(mean), _p(rstd), M, Cdim, _p(s1), _p(s2), DT[dy.dtype], stream_ptr()), "mvlt_bn_bwd_reduce")


def bn_bwd_apply(dy, lddy, z, ldz, mean, rstd, gamma, s1, s2, M, Cdim, dz16, lddz, g_beta=None, g_gamma=None):
    assert dy.dtype in DT and z.dtype in ZDT
    check(L.lib.mvlt_bn_bwd_apply(_p(dy), lddy, _p(z), ldz, ZDT[z.dtype], _p(mean), _p(rstd), _p(gamma), _p(s1), _p(s2), M, Cdim, _p(dz16), lddz,
                                  _p(g_beta), _p(g_gamma), DT[dz16.dtype], DT[dy.dtype], stream_ptr()), "mvlt_bn_bwd_apply")


L.lib.mvlt_ew_mul3_bwd.argtypes = [_vp, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _l, _i, _i, _vp]


def ew_mul3_bwd(dy, lddy, a, b, c, ld, da, db, dc, M, Cdim):
    assert dy.dtype in DT and da.dtype == db.dtype == dc.dtype == dy.dtype and a.dtype == b.dtype == c.dtype and a.dtype in ZDT
    check(L.lib.mvlt_ew_mul3_bwd(_p(dy), lddy, _p(a), _p(b), _p(c), ld, ZDT[a.dtype], _p(da), _p(db), _p(dc), M, Cdim, DT[dy.dtype], stream_ptr()), "mvlt_ew_mul3_bwd")


def ew_mul(out, ldo, a, lda, b, ldb, c=None, ldc=0, *, M, Cdim, accumulate=False, out16=None, ld16=0):
    assert a.dtype == b.dtype and (c is None or c.dtype == a.dtype) and a.dtype in ZDT
    check(L.lib.mvlt_ew_mul(_p(out), ldo, _p(a), lda, _p(b), ldb, _p(c), ldc, ZDT[a.dtype], M, Cdim, 1 if accumulate else 0, _p(out16), ld16,
                            DT[out16.dtype] if out16 is not None else 0, stream_ptr()), "mvlt_ew_mul")


def upsample_fwd(x, ldx, B, H, W, Cdim, scale, out, ldo, nchw=False):
    check(L.lib.mvlt_upsample_fwd(_p(x), ldx, B, H, W, Cdim, scale, _p(out), ldo, DT[out.dtype], 1 if nchw else 0, stream_ptr()), "mvlt_upsample_fwd")


def upsample_bwd(dy, lddy, nchw, B, H, W, Cdim, scale, dx, lddx, accumulate=False):
    assert dy.dtype in DT and dx.dtype in DT
    check(L.lib.mvlt_upsample_bwd(_p(dy), lddy, 1 if nchw else 0, B, H, W, Cdim, scale, _p(dx), lddx, 1 if accumulate else 0, DT[dx.dtype],
                                  DT[dy.dtype], stream_ptr()), "mvlt_upsample_bwd")


L.lib.mvlt_upsample_l1_fwd.argtypes = [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]
L.lib.mvlt_upsample_l1_bwd.argtypes = [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp]


def upsample_l1_ok(W, scale):
    """geometry the fused upsample + SmoothL1 kernels take (the final x8 upsample of the MIM decoder)"""
    return W <= 64 and (W * scale) % 4 == 0 and W * scale <= 256


def upsample_l1_fwd(x, ldx, B, H, W, Cdim, scale, target, loss_sum):
    """loss_sum[0] += sum SmoothL1(upsample(x) - target); x: fp32 [B*H*W, ldx] pixel-major, target: fp32 NCHW [B, C, H*scale, W*scale]"""
    _need_cuda(x, target, loss_sum)
    assert x.dtype == target.dtype == loss_sum.dtype == torch.float32 and target.is_contiguous()
    check(L.lib.mvlt_upsample_l1_fwd(_p(x), ldx, B, H, W, Cdim, scale, _p(target), _p(loss_sum), stream_ptr()), "mvlt_upsample_l1_fwd")


def upsample_l1_bwd(x, ldx, B, H, W, Cdim, scale, target, gscale, dx, lddx):
    """dx[:, :C] = d(mean SmoothL1) / d(x) * gscale[0] (dx bf16 or fp32, row stride lddx)"""
    _need_cuda(x, target, dx)
    assert x.dtype == target.dtype == gscale.dtype == torch.float32 and dx.dtype in DT and target.is_contiguous()
    check(L.lib.mvlt_upsample_l1_bwd(_p(x), ldx, B, H, W, Cdim, scale, _p(target), _p(gscale), _p(dx), lddx, DT[dx.dtype], stream_ptr()),
          "mvlt_upsample_l1_bwd")


# ------------------------------------------------------------------ fused MLP (csrc/mlp.hip), bf16, C in {64, 128}
def mlp_fwd(x, w1, b1, w2, b2, residual, out, M, Cdim, hid, *, row_scale=None, rows_per_scale=0, h_out=None, ln=None, out_op=None, post_ln=None):
    """ln = (gamma, beta, eps, y, mean, rstd): LayerNorm(residual) folded into the operand load -- `x` is then not read (may be None);
    y (bf16 [M, C]) receives the normalised rows, mean / rstd (fp32 [M]) the row statistics.
    out_op: optional bf16 [M, C] copy of the output; `out` (fp32) may be None when only the copy is wanted.
    post_ln = (gamma, beta, eps, y, mean, rstd): LayerNorm of the OUTPUT rows (the next block's norm1) from the epilogue."""
    assert w1.dtype == torch.bfloat16 and w2.dtype == torch.bfloat16
    assert residual.dtype == torch.float32 and residual.is_contiguous() and (out is not None or out_op is not None)
    assert out is None or out.dtype == torch.float32
    assert out_op is None or (out_op.dtype == torch.bfloat16 and out_op.is_contiguous())
    if ln is None:
        assert x.dtype == torch.bfloat16
        tail = (None, None, None, 0.0, None, None, None)
    else:
        g, b, eps, y, mean, rstd = ln
        assert g.dtype == b.dtype == mean.dtype == rstd.dtype == torch.float32 and y.dtype == torch.bfloat16 and y.is_contiguous()
        tail = (ptr(residual), ptr(g), ptr(b), eps, ptr(y), ptr(mean), ptr(rstd))
    post = (None, None, 0.0, None, None, None)
    if post_ln is not None:
        pg, pb, peps, py, pm, pr = post_ln
        assert pg.dtype == pb.dtype == pm.dtype == pr.dtype == torch.float32 and py.dtype == torch.bfloat16 and py.is_contiguous()
        post = (ptr(pg), ptr(pb), peps, ptr(py), ptr(pm), ptr(pr))
    a = L.MlpArgs(ptr(x), None, ptr(w1), ptr(w2), None, ptr(b1), ptr(b2), ptr(residual), ptr(row_scale), rows_per_scale,
                  ptr(out), ptr(h_out), None, None, None, None, M, Cdim, hid, *tail, ptr(out_op), *post)
    check(L.lib.mvlt_mlp_fwd(C.byref(a), stream_ptr()), "mvlt_mlp_fwd")
    return out if out is not None else out_op


def mlp_bwd_dx(x, dy, w1, w1t, w2t, b1, out, M, Cdim, hid, *, row_scale=None, rows_per_scale=0, ln_bwd=None):
    """ln_bwd = dict(x, mean, rstd, gamma, dx, dgamma, dbeta[, dx2, dx2_scale, dx2_rows_per_scale]): the backward of the LayerNorm in
    front of the MLP from this kernel's epilogue -- dx (bf16 [M, C], may be `dy`) += LN backward in place, dx2 = its scaled copy,
    dgamma / dbeta += the column sums (through per-workgroup partials and one mvlt_add_column_sums launch); `out` is not written."""
    assert x.dtype == dy.dtype == torch.bfloat16 and (out is None or out.dtype == torch.bfloat16)
    tail = (None, None, None, 0.0, None, None, None, None, None, None, 0.0, None, None, None)      # ln_* , out_op, post_*
    if ln_bwd is None:
        a = L.MlpArgs(ptr(x), ptr(dy), ptr(w1), ptr(w1t), ptr(w2t), ptr(b1), None, None, ptr(row_scale), rows_per_scale,
                      ptr(out), None, None, None, None, None, M, Cdim, hid, *tail)
        check(L.lib.mvlt_mlp_bwd_dx(C.byref(a), stream_ptr()), "mvlt_mlp_bwd_dx")
        return out
    k = ln_bwd
    assert k["x"].dtype == torch.float32 and k["dx"].dtype == torch.bfloat16 and k["dx"].is_contiguous() and k["x"].is_contiguous()
    assert k["mean"].dtype == k["rstd"].dtype == k["gamma"].dtype == k["dgamma"].dtype == k["dbeta"].dtype == torch.float32
    dx2 = k.get("dx2")
    assert dx2 is None or (dx2.dtype == torch.bfloat16 and dx2.is_contiguous() and k["dx2_scale"].dtype == torch.float32)
    nwg = (M + 127) // 128
    partials = torch.empty(nwg, 2 * Cdim, device=x.device, dtype=torch.float32)
    a = L.MlpArgs(ptr(x), ptr(dy), ptr(w1), ptr(w1t), ptr(w2t), ptr(b1), None, None, ptr(row_scale), rows_per_scale,
                  None, None, None, None, None, None, M, Cdim, hid, *tail,
                  ptr(k["x"]), ptr(k["mean"]), ptr(k["rstd"]), ptr(k["gamma"]), ptr(k["dx"]), ptr(dx2), ptr(k.get("dx2_scale")),
                  int(k.get("dx2_rows_per_scale", 0)), ptr(partials))
    check(L.lib.mvlt_mlp_bwd_dx(C.byref(a), stream_ptr()), "mvlt_mlp_bwd_dx")
    add_column_sums(partials, k["dgamma"], k["dbeta"])
    return k["dx"]


L.lib.mvlt_add_column_sums.argtypes = [_vp, _l, _i, _i, _vp, _i, _vp, _vp]


def add_column_sums(partials, dst0, dst1):
    """dst0 += column sums of partials[:, :n0], dst1 += those of partials[:, n0:] (fp32)"""
    rows, cols = partials.shape
    assert partials.dtype == dst0.dtype == dst1.dtype == torch.float32 and partials.is_contiguous() and dst0.numel() + dst1.numel() == cols
    check(L.lib.mvlt_add_column_sums(_p(partials), rows, cols, cols, _p(dst0), dst0.numel(), _p(dst1), stream_ptr()), "mvlt_add_column_sums")


def mlp_bwd_dw(x, dy, w1, w2t, b1, dw1, db1, dw2, db2, M, Cdim, hid, *, row_scale=None, rows_per_scale=0, partials=None, defer_fold=False):
    """partials = the weight-gradient scratch of gemm_tn (FlatStore.tn_partials()): the token splits leave as bf16 partial tiles + the ordered fold instead of fp32 atomics;
    defer_fold: folded with the other pending ones (tn_fold_flush before anything reads dw1 / dw2)"""
    assert x.dtype == dy.dtype == torch.bfloat16 and dw1.dtype == torch.float32
    a = L.MlpArgs(ptr(x), ptr(dy), ptr(w1), None, ptr(w2t), ptr(b1), None, None, ptr(row_scale), rows_per_scale,
                  None, None, ptr(dw1), ptr(db1), ptr(dw2), ptr(db2), M, Cdim, hid, None, None, None, 0.0, None, None, None)
    if partials is not None:
        a.partials, a.partials_bytes, a.defer_fold = ptr(partials), partials.numel() * partials.element_size(), 1 if defer_fold else 0
    check(L.lib.mvlt_mlp_bwd_dw(C.byref(a), stream_ptr()), "mvlt_mlp_bwd_dw")


# ------------------------------------------------------------------ device-side batch preparation (csrc/batchprep.hip)
_u64 = C.c_uint64
L.lib.mvlt_grid_mask_flags.argtypes = [_vp, _i, _i, _i, _i, _i, _u64, _u64, _vp]
L.lib.mvlt_grid_mask_apply.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]
L.lib.mvlt_token_mask.argtypes = [_vp, _vp, _vp, _i, _i, _u64, _u64, _i, _vp]


def grid_mask_flags(flags, B, gh, gw, num_mask, mode, seed, sample0):
    _need_cuda(flags)
    assert flags.dtype == torch.uint8 and flags.is_contiguous() and flags.numel() == B * gh * gw
    check(L.lib.mvlt_grid_mask_flags(_p(flags), B, gh, gw, num_mask, mode, seed, sample0, stream_ptr()), "mvlt_grid_mask_flags")
    return flags


def grid_mask_apply(image, flags, masked, patch=16, fill=1e-6):
    _need_cuda(image, flags, masked)
    assert image.dtype == torch.float32 and masked.dtype == torch.float32 and image.is_contiguous() and masked.is_contiguous()
    B, Cc, H, W = image.shape
    check(L.lib.mvlt_grid_mask_apply(_p(image), _p(flags), _p(masked), B, Cc, H, W, patch, fill, stream_ptr()), "mvlt_grid_mask_apply")
    return masked


def token_mask(ori_ids, input_ids, labels, seed, sample0, vocab=30522):
    _need_cuda(ori_ids, input_ids, labels)
    assert ori_ids.dtype == input_ids.dtype == labels.dtype == torch.int64 and ori_ids.is_contiguous()
    B, T = ori_ids.shape
    check(L.lib.mvlt_token_mask(_p(ori_ids), _p(input_ids), _p(labels), B, T, seed, sample0, vocab, stream_ptr()), "mvlt_token_mask")


L.lib.mvlt_keep_mask.argtypes = [_vp, _l, _f, _u64, _u64, _vp]
L.lib.mvlt_droppath_scales.argtypes = [_vp, _vp, _i, _i, _u64, _u64, _vp]


def keep_mask(keep, drop_p, seed, call):
    """keep (uint8, any shape, contiguous) <- Bernoulli(1 - drop_p) from Philox(seed; call)"""
    _need_cuda(keep)
    assert keep.dtype == torch.uint8 and keep.is_contiguous()
    check(L.lib.mvlt_keep_mask(_p(keep), keep.numel(), drop_p, seed & (2 ** 64 - 1), call, stream_ptr()), "mvlt_keep_mask")
    return keep


def droppath_scales(out, rates, seed, call):
    """out (fp32 [nrate, ...]) <- Bernoulli(1 - rates[r]) / (1 - rates[r]) per element of row r"""
    _need_cuda(out, rates)
    assert out.dtype == rates.dtype == torch.float32 and out.is_contiguous() and rates.is_contiguous() and out.shape[0] == rates.numel()
    check(L.lib.mvlt_droppath_scales(_p(out), _p(rates), rates.numel(), out.numel() // max(1, rates.numel()), seed & (2 ** 64 - 1), call, stream_ptr()),
          "mvlt_droppath_scales")
    return out


# ------------------------------------------------------------------ position-embedding resize, GELU backward (csrc/elementwise.hip)
L.lib.mvlt_resize_bilinear_tokens.argtypes = [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]
L.lib.mvlt_gelu_bwd.argtypes = [_vp, _vp, _vp, _l, _i, _vp]
L.lib.mvlt_loss_compose.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_float), _vp, _vp, _vp]


def loss_compose(losses, weights, out, total):
    """out[0] = total[0] = sum_i weights[i] * losses[i], out[1 + i] = weights[i] * losses[i]; losses: five fp32 device scalars or None"""
    assert len(losses) == len(weights) == 5 and out.dtype == total.dtype == torch.float32 and out.numel() >= 6 and out.is_contiguous()
    for t in losses:
        assert t is None or (t.is_cuda and t.dtype == torch.float32 and t.numel() == 1)
    ptrs = (C.c_void_p * 5)(*[None if t is None else t.data_ptr() for t in losses])
    ws = (C.c_float * 5)(*[float(w) for w in weights])
    check(L.lib.mvlt_loss_compose(ptrs, ws, _p(out), _p(total), stream_ptr()), "mvlt_loss_compose")
    return out


L.lib.mvlt_resize_bilinear_tokens_multi.argtypes = [_vp] * 9 + [_i, _i, _vp]


def resize_bilinear_tokens_multi(jobs, adjoint=False):
    """jobs: up to four (src, dst, hin, win, hout, wout, Cdim) like resize_bilinear_tokens, one launch"""
    n = len(jobs)
    assert 1 <= n <= 4
    for src, dst, *_ in jobs:
        assert src.dtype == torch.float32 and dst.dtype == torch.float32 and src.stride(-1) == 1 and dst.stride(-1) == 1
    P, I = C.c_void_p * n, C.c_int * n
    arr = lambda f: I(*[f(j) for j in jobs])
    check(L.lib.mvlt_resize_bilinear_tokens_multi(P(*[j[0].data_ptr() for j in jobs]), arr(lambda j: j[0].stride(0)), P(*[j[1].data_ptr() for j in jobs]),
                                                  arr(lambda j: j[1].stride(0)), arr(lambda j: j[2]), arr(lambda j: j[3]), arr(lambda j: j[4]), arr(lambda j: j[5]),
                                                  arr(lambda j: j[6]), n, 1 if adjoint else 0, stream_ptr()), "mvlt_resize_bilinear_tokens_multi")


def resize_bilinear_tokens(src, dst, hin, win, hout, wout, Cdim, adjoint=False):
    """src [hin*win, C] -> dst [hout*wout, C] (fp32, rows Cdim floats apart); adjoint: src is d(dst-shaped), accumulated into dst = d(source map)"""
    _need_cuda(src, dst)
    assert src.dtype == torch.float32 and dst.dtype == torch.float32 and src.stride(-1) == 1 and dst.stride(-1) == 1
    check(L.lib.mvlt_resize_bilinear_tokens(_p(src), src.stride(0), _p(dst), dst.stride(0), hin, win, hout, wout, Cdim, 1 if adjoint else 0, stream_ptr()),
          "mvlt_resize_bilinear_tokens")
    return dst


def gelu_bwd(dy, h, out):
    _need_cuda(dy, h, out)
    assert dy.dtype == h.dtype == out.dtype and dy.is_contiguous() and h.is_contiguous() and out.is_contiguous()
    check(L.lib.mvlt_gelu_bwd(_p(dy), _p(h), _p(out), dy.numel(), DT[dy.dtype], stream_ptr()), "mvlt_gelu_bwd")
    return out
