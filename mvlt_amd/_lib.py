"""ctypes binding of libmvlt_hip.so (the C ABI declared in include/mvlt_hip.h).

The product path has NO fallback: importing this module without the built library, or calling an op without a
GPU, raises.  torch is used only for device memory and streams.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MVLT_HIP_LIB") or os.path.join(_HERE, "libmvlt_hip.so")      # override: A/B runs of two builds


class MVLTError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} is missing: build it with `python -m mvlt_amd.build` (hipcc, gfx950). "
                      "mvlt_amd has no CPU or eager fallback.")
lib = C.CDLL(LIB_PATH)

c_int, c_float, c_void_p, c_long = C.c_int, C.c_float, C.c_void_p, C.c_long


class RowMap(C.Structure):
    _fields_ = [(n, c_int) for n in ("mode", "rows_per_batch", "batch_stride", "offset",
                                     "r", "w_in", "tokens_in", "hw_out", "w_out", "c_seg", "h_in")]


class GemmNTArgs(C.Structure):
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("C", c_void_p),
                ("M", c_int), ("N", c_int), ("K", c_int), ("lda", c_int), ("ldb", c_int), ("ldc", c_int),
                ("dtype", c_int), ("out_dtype", c_int),
                ("a_map", RowMap), ("c_map", RowMap),
                ("bias", c_void_p), ("act", c_int), ("H", c_void_p),
                ("row_scale", c_void_p), ("rows_per_scale", c_int), ("R", c_void_p),
                ("col_sum", c_void_p), ("col_sumsq", c_void_p), ("col_copies", c_int), ("split_k", c_int),
                ("post_y", c_void_p), ("post_ld", c_int), ("post_gamma", c_void_p), ("post_beta", c_void_p), ("post_eps", C.c_float),
                ("post_mean", c_void_p), ("post_rstd", c_void_p), ("r_fp32", c_int)]


class PrepDesc(C.Structure):
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("kind", c_int), ("R", c_int), ("C", c_int), ("ld_out", c_int),
                ("d0", c_int), ("d1", c_int), ("d2", c_int), ("src_off", c_int), ("ss0", c_int), ("ss1", c_int), ("ss2", c_int),
                ("ds0", c_int), ("ds1", c_int), ("ds2", c_int)]


class GemmTNArgs(C.Structure):
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("C", c_void_p),
                ("M", c_int), ("N1", c_int), ("N2", c_int), ("lda", c_int), ("ldb", c_int), ("ldc", c_int),
                ("dtype", c_int), ("a_map", RowMap), ("b_map", RowMap),
                ("colsum_a", c_void_p), ("splits", c_int), ("colsum_b", c_void_p), ("trans_c", c_int),
                ("c_taps", c_int), ("c_seg", c_int), ("dgrad_wt", c_void_p), ("dgrad_out", c_void_p), ("dgrad_ld", c_int),
                ("partials", c_void_p), ("partials_bytes", c_long), ("defer_fold", c_int), ("c_overwrite", c_int)]


class LayerNormArgs(C.Structure):
    _fields_ = [("x", c_void_p), ("y", c_void_p), ("gamma", c_void_p), ("beta", c_void_p),
                ("mean", c_void_p), ("rstd", c_void_p), ("add", c_void_p), ("add_rows", c_int),
                ("rows", c_int), ("C", c_int), ("ldx", c_int), ("ldy", c_int),
                ("x_map", RowMap), ("y_map", RowMap), ("eps", c_float), ("dtype", c_int), ("y_dtype", c_int),
                ("y2", c_void_p), ("gamma2", c_void_p), ("beta2", c_void_p), ("eps2", c_float), ("mean2", c_void_p), ("rstd2", c_void_p)]


class LayerNormBwdArgs(C.Structure):
    _fields_ = [("dy", c_void_p), ("x", c_void_p), ("dx", c_void_p),
                ("gamma", c_void_p), ("mean", c_void_p), ("rstd", c_void_p),
                ("dgamma", c_void_p), ("dbeta", c_void_p),
                ("rows", c_int), ("C", c_int), ("lddy", c_int), ("ldx", c_int), ("lddx", c_int),
                ("dy_map", RowMap), ("x_map", RowMap), ("dx_map", RowMap),
                ("dx_accumulate", c_int), ("dtype", c_int), ("x_dtype", c_int), ("dx_dtype", c_int),
                ("dx2", c_void_p), ("dx2_scale", c_void_p), ("dx2_rows_per_scale", c_int), ("lddx2", c_int),
                ("dg_copies", c_int), ("dg_copy_stride", C.c_long)]


class AttnArgs(C.Structure):
    _fields_ = [("Q", c_void_p), ("KV", c_void_p), ("O", c_void_p), ("lse", c_void_p),
                ("B", c_int), ("H", c_int), ("N", c_int), ("M", c_int),
                ("ldq", c_int), ("ldkv", c_int), ("ldo", c_int), ("k_off", c_int), ("v_off", c_int),
                ("scale", c_float), ("dtype", c_int)]


class AttnBwdArgs(C.Structure):
    _fields_ = [("Q", c_void_p), ("KV", c_void_p), ("O", c_void_p), ("dO", c_void_p), ("lse", c_void_p),
                ("dQ", c_void_p), ("dKV", c_void_p),
                ("B", c_int), ("H", c_int), ("N", c_int), ("M", c_int),
                ("ldq", c_int), ("ldkv", c_int), ("ldo", c_int), ("lddkv", c_int),
                ("k_off", c_int), ("v_off", c_int), ("scale", c_float), ("dtype", c_int), ("dkv_dtype", c_int)]


class MlpArgs(C.Structure):
    _fields_ = [("x", c_void_p), ("dy", c_void_p), ("w1", c_void_p), ("wb", c_void_p), ("wc", c_void_p),
                ("b1", c_void_p), ("b2", c_void_p), ("residual", c_void_p), ("row_scale", c_void_p), ("rows_per_scale", c_int),
                ("out", c_void_p), ("h_out", c_void_p), ("dw1", c_void_p), ("db1", c_void_p), ("dw2", c_void_p), ("db2", c_void_p),
                ("M", c_int), ("C", c_int), ("hid", c_int),
                ("ln_x", c_void_p), ("ln_gamma", c_void_p), ("ln_beta", c_void_p), ("ln_eps", c_float),
                ("ln_y", c_void_p), ("ln_mean", c_void_p), ("ln_rstd", c_void_p), ("out_op", c_void_p),
                ("post_gamma", c_void_p), ("post_beta", c_void_p), ("post_eps", c_float),
                ("post_y", c_void_p), ("post_mean", c_void_p), ("post_rstd", c_void_p),
                ("lnb_x", c_void_p), ("lnb_mean", c_void_p), ("lnb_rstd", c_void_p), ("lnb_gamma", c_void_p),
                ("lnb_dx", c_void_p), ("lnb_dx2", c_void_p), ("lnb_dx2_scale", c_void_p), ("lnb_dx2_rows_per_scale", c_int),
                ("lnb_partials", c_void_p), ("partials", c_void_p), ("partials_bytes", c_long), ("defer_fold", c_int)]


lib.mvlt_last_error.restype = C.c_char_p
lib.mvlt_last_kernel.restype = C.c_char_p
lib.mvlt_sizeof.argtypes = [C.c_char_p]
ABI_VERSION = 6          # include/mvlt_hip.h MVLT_ABI_VERSION this binding was written against
if lib.mvlt_abi_version() != ABI_VERSION:
    raise ImportError(f"ABI mismatch: {LIB_PATH} is version {lib.mvlt_abi_version()}, the binding is version {ABI_VERSION} (stale build? run python -m mvlt_amd.build)")
for _name, _cls in (("mvlt_rowmap", RowMap), ("mvlt_prep_desc", PrepDesc), ("mvlt_gemm_nt_args", GemmNTArgs), ("mvlt_gemm_tn_args", GemmTNArgs),
                    ("mvlt_layernorm_args", LayerNormArgs), ("mvlt_layernorm_bwd_args", LayerNormBwdArgs),
                    ("mvlt_attn_args", AttnArgs), ("mvlt_attn_bwd_args", AttnBwdArgs), ("mvlt_mlp_args", MlpArgs)):
    _n = lib.mvlt_sizeof(_name.encode())
    if _n != C.sizeof(_cls):
        raise ImportError(f"ABI mismatch for {_name}: library says {_n} bytes, binding has {C.sizeof(_cls)}")

EXPORTS = ["mvlt_last_error", "mvlt_last_kernel", "mvlt_abi_version", "mvlt_sizeof", "mvlt_gemm_nt", "mvlt_gemm_tn",
           "mvlt_layernorm_fwd", "mvlt_layernorm_bwd", "mvlt_fold_copies", "mvlt_batch_sum", "mvlt_sr_attention_fwd", "mvlt_sr_attention_bwd",
           "mvlt_bert_embed_fwd", "mvlt_bert_embed_bwd", "mvlt_patchify", "mvlt_masked_select", "mvlt_gather_rows",
           "mvlt_scatter_rows", "mvlt_cross_entropy_fwd", "mvlt_cross_entropy_bwd", "mvlt_adamw_step", "mvlt_smooth_l1_fwd", "mvlt_smooth_l1_bwd", "mvlt_cast_bf16",
           "mvlt_transpose_cast", "mvlt_row_scale", "mvlt_head_grad_prep", "mvlt_weight_prep", "mvlt_col_stats", "mvlt_bn_finalize", "mvlt_bn_finalize_norm", "mvlt_bn_norm", "mvlt_bn_bwd_reduce", "mvlt_bn_bwd_apply", "mvlt_ew_mul3_bwd",
           "mvlt_ew_mul", "mvlt_upsample_fwd", "mvlt_upsample_bwd", "mvlt_mlp_fwd", "mvlt_mlp_bwd_dx", "mvlt_mlp_bwd_dw",
           "mvlt_grid_mask_flags", "mvlt_grid_mask_apply", "mvlt_token_mask", "mvlt_resize_bilinear_tokens", "mvlt_resize_bilinear_tokens_multi", "mvlt_gelu_bwd",
           "mvlt_keep_mask", "mvlt_droppath_scales", "mvlt_loss_compose", "mvlt_add_column_sums",
           "mvlt_upsample_l1_fwd", "mvlt_upsample_l1_bwd", "mvlt_tn_fold_flush", "mvlt_tn_fold_discard", "mvlt_sr_attention_bwd_chunks"]

DT = {torch.bfloat16: 0, torch.float32: 1}


def last_kernel():
    """name of the kernel instantiation launched last on this thread, e.g. "mlp_wgrad2_kernel<64, 4>" (the runtime's name for the launched
    function, demangled; return type, namespace and parameter list stripped)"""
    s = lib.mvlt_last_kernel().decode()
    s = s.replace("(anonymous namespace)::", "")
    if s.startswith("void "):
        s = s[5:]
    depth = 0
    for i, ch in enumerate(s):                   # cut at the '(' that opens the parameter list (outside the template brackets)
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            return s[:i]
    return s


def check(rc, what):
    if rc != 0:
        raise MVLTError(f"{what} failed ({rc}): {lib.mvlt_last_error().decode()}")


_raw_stream, _cur_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice


def stream_ptr():
    """torch's current stream of the current device as a hipStream_t (two C calls: `torch.cuda.current_stream().cuda_stream` walks ~10 Python frames and
    builds a Stream object -- 358 launches per step made that 2.5 ms of a 10 ms host step, tools/host_profile.py)"""
    return c_void_p(_raw_stream(_cur_device()))


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise MVLTError("mvlt_amd ops need CUDA/HIP tensors (no CPU fallback)")
    return c_void_p(t.data_ptr())


def rowmap(rows_per_batch=0, batch_stride=0, offset=0):
    return RowMap(0, rows_per_batch, batch_stride, offset, 0, 0, 0, 0, 0, 0, 0)


def patchmap(r, w_in, tokens_in, hw_out, w_out, c_seg):
    return RowMap(1, 0, 0, 0, r, w_in, tokens_in, hw_out, w_out, c_seg, 0)


def conv3map(h, w, tokens_in, c_seg):
    """3x3 / pad 1 neighbourhood gather over an h x w pixel grid stored pixel-major with `tokens_in` rows per batch"""
    return RowMap(2, 0, 0, 0, 3, w, tokens_in, h * w, w, c_seg, h)
