"""MI355X-native PVLT (Pyramid Vision-Language Transformer) behind the reference's constructor API.

Drop-in counterpart of reference libs/pvlt.py: `pvlt_tiny/small/medium/large(pretrained, token_hidden_size,
num_text_tokens, loss_type, pretrained_pth, **kwargs)` return an nn.Module whose `state_dict()` has the reference's
keys/shapes and whose `forward(input_images, input_ids)` returns the reference's logits dict
(libs/pvlt.py:358-401).  Nothing here is a translation of the reference module code: parameters live in one flat
fp32 buffer, activations are token-major (B, HW+T, C) in the compute dtype for the whole trunk (no NCHW round
trips, no torch.cat/split), and forward/backward are explicit schedules of the HIP kernels in mvlt_amd/csrc through
the C ABI (include/mvlt_hip.h).  There is no CPU or eager fallback: without libmvlt_hip.so or a GPU this raises.

Extra entry point (used by engine_grid_masking.train_one_epoch_vl): `forward(images, ids, mlm_labels=...)` runs
the MLM head only on the rows CrossEntropyLoss(ignore_index=-1) would keep (masked-index selection) and returns
the summed/averaged loss directly instead of the (B, T, 30522) logits.
"""
import math
from functools import partial

import torch
import torch.nn as nn

from . import ops
from ._lib import patchmap, rowmap
from .params import FlatStore, Holder, affine, conv_default_init_, linear, trunc_normal_

__all__ = ["pvlt_tiny", "pvlt_small", "pvlt_medium", "pvlt_large", "PyramidVisionLanguageTransformer"]

VOCAB = 30522
VOCAB_LD = 30528                      # vocabulary rows padded to 16 B for the logits / dlogits buffers
EPS_BLOCK, EPS_DEFAULT, EPS_BERT = 1e-6, 1e-5, 1e-12
BERT_DROP = 0.1


# =============================================================================================== parameter tree
class _Attn(nn.Module):
    def __init__(self, C, sr):
        super().__init__()
        self.q = linear(C, C)
        self.kv = linear(2 * C, C)
        self.proj = linear(C, C)
        if sr > 1:
            self.sr = Holder(weight=(C, C, sr, sr), bias=(C,))
            self.norm = affine(C)


class _Mlp(nn.Module):
    def __init__(self, C, hid):
        super().__init__()
        self.fc1 = linear(hid, C)
        self.fc2 = linear(C, hid)


class _Block(nn.Module):
    def __init__(self, C, sr, hid):
        super().__init__()
        self.norm1 = affine(C)
        self.attn = _Attn(C, sr)
        self.norm2 = affine(C)
        self.mlp = _Mlp(C, hid)


class _PatchEmbed(nn.Module):
    def __init__(self, cin, C, k):
        super().__init__()
        self.proj = Holder(weight=(C, cin, k, k), bias=(C,))
        self.norm = affine(C)


class _BertEmbeddings(nn.Module):
    def __init__(self, hidden):
        super().__init__()
        self.word_embeddings = Holder(weight=(VOCAB, hidden))
        self.position_embeddings = Holder(weight=(512, hidden))
        self.token_type_embeddings = Holder(weight=(2, hidden))
        self.LayerNorm = affine(hidden)


class _Transform(nn.Module):
    def __init__(self, hidden):
        super().__init__()
        self.dense = linear(hidden, hidden)
        self.LayerNorm = affine(hidden)


class _MLMHead(nn.Module):
    def __init__(self, hidden, tied_weight):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(VOCAB))
        self.transform = _Transform(hidden)
        self.mlm_decoder = nn.Module()
        self.mlm_decoder.weight = tied_weight          # same Parameter object as the word embedding table


class _ClsHead(nn.Module):
    def __init__(self, hidden, n):
        super().__init__()
        self.linear_bias = nn.Parameter(torch.zeros(n))
        self.linear = linear(n, hidden)


def _conv_bn(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout))


class _ITGHead(nn.Module):
    """MIM decoder parameters (reference libs/vl_heads.py:107-134), evaluated by the HIP schedule in mvlt_amd/mim.py (the same graph
    on PyTorch-ROCm ops, the A/B reference of the HIP-vs-torch test, lives in tests/mim_twin.py)."""

    def __init__(self, dims, ch=64):
        super().__init__()
        self.reduction1 = _conv_bn(dims[1], ch)
        self.reduction2 = _conv_bn(dims[2], ch)
        self.reduction3 = _conv_bn(dims[3], ch)
        self.conv_upsample1 = _conv_bn(ch, ch)
        self.conv_upsample2 = _conv_bn(ch, ch)
        self.conv_upsample3 = _conv_bn(ch, ch)
        self.conv_upsample4 = _conv_bn(ch, ch)
        self.conv_upsample5 = _conv_bn(2 * ch, 2 * ch)
        self.conv_concat2 = _conv_bn(2 * ch, 2 * ch)
        self.conv_concat3 = _conv_bn(3 * ch, 3 * ch)
        self.conv4 = _conv_bn(3 * ch, 3 * ch)
        self.score = nn.Sequential(nn.Conv2d(3 * ch, 3, 1))


# =============================================================================================== the model
class PyramidVisionLanguageTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dims=(64, 128, 256, 512),
                 num_heads=(1, 2, 4, 8), mlp_ratios=(4, 4, 4, 4), qkv_bias=False, qk_scale=None, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0., norm_layer=None, depths=(3, 4, 6, 3), sr_ratios=(8, 4, 2, 1),
                 num_stages=4, F4=False, token_hidden_size=768, num_text_tokens=128,
                 loss_type=None, compute_dtype=torch.bfloat16, **unused):
        super().__init__()
        loss_type = loss_type if loss_type is not None else {'itm': 1, 'mlm': 1, 'itg': 1, 'rtd': 1}
        assert num_stages == 4, "PVLT has four stages"
        assert qkv_bias, "the HIP schedule assumes qkv_bias=True (every reference factory sets it)"
        assert drop_rate == 0.0 and attn_drop_rate == 0.0, "drop_rate / attn_drop_rate are 0 in every reference config"
        assert token_hidden_size == 768, "BertEmbeddings hidden size is 768 (bert-base-uncased)"
        for d, h in zip(embed_dims, num_heads):
            assert d % h == 0, f"dim {d} should be divided by num_heads {h}."
            assert d // h == 64, "head_dim must be 64 (all reference variants)"
        assert img_size % patch_size == 0, f"img_size {img_size} should be divided by patch_size {patch_size}."
        self.num_classes, self.depths, self.F4, self.num_stages = num_classes, tuple(depths), F4, num_stages
        self.T_num = num_text_tokens
        self.loss_type = loss_type
        self.dims, self.heads, self.sr = tuple(embed_dims), tuple(num_heads), tuple(sr_ratios)
        self.hid = tuple(int(d * r) for d, r in zip(embed_dims, mlp_ratios))
        self.patch_size, self.in_chans, self.ctor_img_size = patch_size, in_chans, img_size
        self.hidden = token_hidden_size
        self.compute_dtype = compute_dtype
        self.dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]

        for i in range(4):
            size_i = img_size if i == 0 else img_size // (2 ** (i + 1))
            k = patch_size if i == 0 else 2
            assert size_i % k == 0, f"img_size {size_i} should be divided by patch_size {k}."
            grid = size_i // k
            n_pos = grid * grid + (1 if i == 3 else 0)
            C = embed_dims[i]
            setattr(self, f"patch_embed{i+1}", _PatchEmbed(in_chans if i == 0 else embed_dims[i - 1], C, k))
            setattr(self, f"text_embed{i+1}", nn.Sequential(linear(C, token_hidden_size if i == 0 else embed_dims[i - 1]), affine(C)))
            setattr(self, f"pos_embed{i+1}", nn.Parameter(torch.zeros(1, n_pos, C)))
            setattr(self, f"text_pos_embed{i+1}", nn.Parameter(torch.zeros(1, num_text_tokens, C)))
            setattr(self, f"block{i+1}", nn.ModuleList([_Block(C, sr_ratios[i], self.hid[i]) for _ in range(depths[i])]))
        self.grids = [img_size // patch_size // (2 ** i) for i in range(4)]

        self.text_embeddings = _BertEmbeddings(token_hidden_size)
        Hd = token_hidden_size
        if loss_type['mlm'] == 1:
            self.mlm_head_embed = nn.Sequential(linear(Hd, embed_dims[-1]), affine(Hd))
            self.mlm_head = _MLMHead(Hd, self.text_embeddings.word_embeddings.weight)
        if loss_type['itm'] == 1:
            self.itm_head_embed = nn.Sequential(linear(Hd, embed_dims[-1]), affine(Hd))
            self.itm_head = _ClsHead(Hd, 2)
        if loss_type['cls'] == 1:
            self.sup_cls_head_embed = nn.Sequential(linear(Hd, embed_dims[-1]), affine(Hd))
            self.sup_cls_head = _ClsHead(Hd, 48)
            self.sub_cls_head_embed = nn.Sequential(linear(Hd, embed_dims[-1]), affine(Hd))
            self.sub_cls_head = _ClsHead(Hd, 122)
        if loss_type['t2i'] == 1:
            self.t2i_head = _ITGHead(embed_dims, 64)
        self._init_weights()
        self.register_load_state_dict_pre_hook(self._drop_legacy_keys)

        self._store = FlatStore(self, compute_dtype)
        self._anchor = None
        self._transposed, self._conv_perm = self._operand_lists()
        self._conv3 = [n for n, p in self.named_parameters() if n.startswith("t2i_head.") and p.dim() == 4 and p.shape[-1] == 3]
        self.injected_masks = None      # tests: {'bert': (B,T,768) keep, 'droppath': [...], 'droppath2': [...]}

    # ------------------------------------------------------------------ init / state
    def _init_weights(self):
        """Same distributions as reference libs/pvlt.py:228-229,280-289 (+ PyTorch defaults where it keeps them)."""
        for name, p in self.named_parameters():
            if name.startswith("t2i_head"):
                continue                                    # real nn.Conv2d / BatchNorm2d: PyTorch defaults
            if "pos_embed" in name:
                trunc_normal_(p, std=.02)
            elif name.endswith("position_embeddings.weight") or name.endswith("token_type_embeddings.weight"):
                nn.init.normal_(p, 0.0, 1.0)               # nn.Embedding default (not touched by _init_weights)
            elif name.endswith("word_embeddings.weight"):
                trunc_normal_(p, std=.02)                   # re-initialised through the tied nn.Linear decoder
            elif p.dim() == 4:                              # patch-embed / sr convs keep nn.Conv2d defaults
                conv_default_init_(p, None)
            elif p.dim() == 2:
                trunc_normal_(p, std=.02)
            elif name.endswith("norm.weight") or name.endswith("norm1.weight") or name.endswith("norm2.weight") \
                    or name.endswith("LayerNorm.weight") or name.endswith(".1.weight"):
                nn.init.ones_(p)
            else:
                nn.init.zeros_(p)                           # biases, LN shifts, mlm bias, linear_bias
        for i in range(4):                                   # conv biases: nn.Conv2d default U(-1/sqrt(fan_in), ..)
            pe = getattr(self, f"patch_embed{i+1}").proj
            conv_default_init_(pe.weight, pe.bias)
            for blk in getattr(self, f"block{i+1}"):
                if hasattr(blk.attn, "sr"):
                    conv_default_init_(blk.attn.sr.weight, blk.attn.sr.bias)

    @staticmethod
    def _drop_legacy_keys(module, state_dict, prefix, *args):
        # transformers==4.10.2 checkpoints carry a persistent BertEmbeddings.position_ids buffer
        state_dict.pop(prefix + "text_embeddings.position_ids", None)

    def _operand_lists(self):
        transposed, conv_perm = [], []
        for name, p in self.named_parameters():
            if name.startswith("t2i_head") or "embeddings.position" in name or "token_type" in name:
                continue
            if p.dim() == 2 and "pos_embed" not in name and name != "mlm_head.mlm_decoder.weight":
                transposed.append(name)
            if p.dim() == 4 and not name.startswith("patch_embed1."):
                conv_perm.append(name)
        return transposed, conv_perm

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        if hasattr(self, "_store"):
            self._store.P = None                            # parameter storage was replaced: rebuild lazily
        return out

    @property
    def store(self):
        return self._store

    def set_compute_dtype(self, dtype):
        """bf16 (default) or fp32 (exact-f32 MFMA path; the reference's `fp32=True` / --fp32-resume switch)."""
        assert dtype in (torch.bfloat16, torch.float32)
        if dtype != self.compute_dtype:
            self.compute_dtype = dtype
            self._store.compute_dtype = dtype
            self._store.P = None

    # ------------------------------------------------------------------ forward
    def forward(self, input_images, input_ids, mlm_labels=None, mlm_positions=None, mlm_count=None, t2i_target=None):
        if not input_images.is_cuda:
            raise RuntimeError("mvlt_amd PVLT runs on MI355X only (HIP kernels); there is no CPU path. "
                               "Use oracle/pvlt_oracle.py for CPU checks.")
        from .schedule import run_forward
        return run_forward(self, input_images, input_ids, mlm_labels, mlm_positions, mlm_count, t2i_target)


    def forward_pyramid_features_vl(self, x, y):
        """(img_feats, text_feats) of the four stages like the reference method of the same name (libs/pvlt.py:322-356): image
        features (B, C_i, H_i, W_i) contiguous, text features (B, T, C_i).  The stage outputs are the trunk's token buffers; the
        two views per stage are made by ATen (this is an API-shape adapter for callers outside the hot path -- `forward` hands the
        token buffers to the heads directly)."""
        if not x.is_cuda:
            raise RuntimeError("mvlt_amd PVLT runs on MI355X only (HIP kernels); there is no CPU path.")
        from .schedule import run_trunk
        outs = run_trunk(self, x, y)
        B, side = x.shape[0], x.shape[2] // self.patch_size
        img_feats, text_feats = [], []
        for i, t in enumerate(outs):
            s_i = side // (2 ** i)
            img_feats.append(t[:, : s_i * s_i, :].reshape(B, s_i, s_i, -1).permute(0, 3, 1, 2).contiguous())
            text_feats.append(t[:, s_i * s_i:, :])
        return img_feats, text_feats


def _cfg(url='', **kwargs):     # timm.models.vision_transformer._cfg metadata (stored as model.default_cfg, unused)
    return {'url': url, 'num_classes': 1000, 'input_size': (3, 224, 224), 'pool_size': None, 'crop_pct': .9,
            'interpolation': 'bicubic', 'mean': (0.485, 0.456, 0.406), 'std': (0.229, 0.224, 0.225),
            'first_conv': 'patch_embed.proj', 'classifier': 'head', **kwargs}


def _factory(depths, label):
    def make(pretrained=False, token_hidden_size=768, num_text_tokens=128, loss_type=None, pretrained_pth=None, **kwargs):
        kwargs.pop("drop_block_rate", None)
        model = PyramidVisionLanguageTransformer(
            patch_size=4, embed_dims=[64, 128, 320, 512], num_heads=[1, 2, 5, 8], mlp_ratios=[8, 8, 4, 4], qkv_bias=True,
            norm_layer=partial(nn.LayerNorm, eps=1e-6), depths=depths, sr_ratios=[8, 4, 2, 1],
            token_hidden_size=token_hidden_size, num_text_tokens=num_text_tokens, loss_type=loss_type, **kwargs)
        model.default_cfg = _cfg()
        if pretrained_pth:
            model.load_state_dict(torch.load(pretrained_pth, map_location="cpu"), strict=False)
            print('>>> load pretrained weights (backbone part) from:', pretrained_pth)
        return model
    make.__name__ = label
    make.__doc__ = f"{label}: reference libs/pvlt.py factory of the same name (depths {depths})."
    return make


pvlt_tiny = _factory([2, 2, 2, 2], "pvlt_tiny")
pvlt_small = _factory([3, 4, 6, 3], "pvlt_small")
pvlt_medium = _factory([3, 4, 18, 3], "pvlt_medium")
pvlt_large = _factory([3, 8, 27, 3], "pvlt_large")

try:    # main_vl.py:259-270 builds the model with timm.models.create_model(args.model, ...): the import of this module registers the factories when timm exists
    from timm.models.registry import register_model as _register_model
except ImportError:   # timm is not installed in this image (tests/test_host_cpu.py::test_timm_create_model_path runs the registration against a timm-0.3.2-shaped registry)
    _register_model = None
if _register_model is not None:       # a registry that is there and refuses the factories is an error, not something to swallow
    for _f in (pvlt_tiny, pvlt_small, pvlt_medium, pvlt_large):
        _register_model(_f)
