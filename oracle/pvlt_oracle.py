"""CPU oracle for the MVLT hot path (TEST INFRASTRUCTURE -- never shipped, never timed
as the product).

A functional, plain-PyTorch fp32 restatement of the reference's model forward
and train-step loss.  It operates on a flat ``state_dict``-style mapping
``name -> tensor`` whose names/shapes are the reference's,
so the same synthetic weights (``oracle/filler.py``) drive the reference, this
oracle and the HIP product.

Parity pin: ``tests/golden/make_golden.py`` imports the real reference in the
build container, runs it on filler weights/inputs and checks this oracle
against it (<=1e-5) before writing ``tests/golden/*.npz``; the CPU test-suite
re-checks the oracle against those committed fixtures.

Reference lines followed (all under /root/reference):
  trunk           libs/pvlt.py:322-356      heads   libs/pvlt.py:358-401
  Attention       libs/pvlt.py:95-121       Mlp     libs/pvlt.py:65-71
  Block           libs/pvlt.py:140-144      PatchEmbed libs/pvlt.py:165-172
  pos-embed       libs/pvlt.py:291-297      MLM/ITM/CLS/ITG heads libs/vl_heads.py:13-165
  loss step       engine_grid_masking.py:69-102
  BertEmbeddings  transformers==4.10.2 (not vendored; call sites libs/pvlt.py:232-233,326)
  DropPath        timm==0.3.2 (not vendored; call site libs/pvlt.py:135,141-142)
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

VARIANT_DEPTHS = {  # libs/pvlt.py:420,438,455,473
    "pvlt_tiny": (2, 2, 2, 2),
    "pvlt_small": (3, 4, 6, 3),
    "pvlt_medium": (3, 4, 18, 3),
    "pvlt_large": (3, 8, 27, 3),
}
DIMS = (64, 128, 320, 512)       # libs/pvlt.py:419
HEADS = (1, 2, 5, 8)
MLP_RATIOS = (8, 8, 4, 4)
SR = (8, 4, 2, 1)
VOCAB = 30522
N_SUP, N_SUB = 48, 122           # libs/pvlt.py:267,273
EPS_BLOCK, EPS_DEFAULT, EPS_BERT = 1e-6, 1e-5, 1e-12


class Cfg:
    def __init__(self, variant="pvlt_tiny", loss_type=None, ctor_img_size=224, token_hidden=768,
                 num_text_tokens=128, drop_path_rate=0.0, in_chans=3):
        self.variant = variant
        self.depths = VARIANT_DEPTHS[variant]
        self.loss_type = dict(loss_type or {"mlm": 1, "itm": 1, "t2i": 1, "cls": 0})
        self.ctor_img_size = ctor_img_size      # pos-embeds are sized for THIS (224 by default)
        self.token_hidden = token_hidden
        self.T = num_text_tokens
        self.in_chans = in_chans
        n = sum(self.depths)
        # torch.linspace(0, r, n) evaluated in fp32 like libs/pvlt.py:197
        self.dpr = [v.item() for v in torch.linspace(0, drop_path_rate, n)]

    def grid(self, stage):        # constructor-time patch grid of stage (0-based)
        return self.ctor_img_size // (4 * 2 ** stage)


def param_shapes(cfg, bert_position_ids=False):
    """Ordered name->shape for the reference's state_dict (tied decoder listed too)."""
    s = OrderedDict()
    for i in range(4):
        g = cfg.grid(i)
        n_pos = g * g + (1 if i == 3 else 0)
        s[f"pos_embed{i+1}"] = (1, n_pos, DIMS[i])
        s[f"text_pos_embed{i+1}"] = (1, cfg.T, DIMS[i])
    for i in range(4):
        C = DIMS[i]
        cin = cfg.in_chans if i == 0 else DIMS[i - 1]
        k = 4 if i == 0 else 2
        s[f"patch_embed{i+1}.proj.weight"] = (C, cin, k, k)
        s[f"patch_embed{i+1}.proj.bias"] = (C,)
        s[f"patch_embed{i+1}.norm.weight"] = (C,)
        s[f"patch_embed{i+1}.norm.bias"] = (C,)
        tin = cfg.token_hidden if i == 0 else DIMS[i - 1]
        s[f"text_embed{i+1}.0.weight"] = (C, tin)
        s[f"text_embed{i+1}.0.bias"] = (C,)
        s[f"text_embed{i+1}.1.weight"] = (C,)
        s[f"text_embed{i+1}.1.bias"] = (C,)
        for j in range(cfg.depths[i]):
            p = f"block{i+1}.{j}."
            s[p + "norm1.weight"] = (C,)
            s[p + "norm1.bias"] = (C,)
            s[p + "attn.q.weight"] = (C, C)
            s[p + "attn.q.bias"] = (C,)
            s[p + "attn.kv.weight"] = (2 * C, C)
            s[p + "attn.kv.bias"] = (2 * C,)
            s[p + "attn.proj.weight"] = (C, C)
            s[p + "attn.proj.bias"] = (C,)
            if SR[i] > 1:
                s[p + "attn.sr.weight"] = (C, C, SR[i], SR[i])
                s[p + "attn.sr.bias"] = (C,)
                s[p + "attn.norm.weight"] = (C,)
                s[p + "attn.norm.bias"] = (C,)
            s[p + "norm2.weight"] = (C,)
            s[p + "norm2.bias"] = (C,)
            hid = C * MLP_RATIOS[i]
            s[p + "mlp.fc1.weight"] = (hid, C)
            s[p + "mlp.fc1.bias"] = (hid,)
            s[p + "mlp.fc2.weight"] = (C, hid)
            s[p + "mlp.fc2.bias"] = (C,)
    Hd = cfg.token_hidden
    if bert_position_ids:
        s["text_embeddings.position_ids"] = (1, 512)
    s["text_embeddings.word_embeddings.weight"] = (VOCAB, Hd)
    s["text_embeddings.position_embeddings.weight"] = (512, Hd)
    s["text_embeddings.token_type_embeddings.weight"] = (2, Hd)
    s["text_embeddings.LayerNorm.weight"] = (Hd,)
    s["text_embeddings.LayerNorm.bias"] = (Hd,)

    def embed(prefix):
        s[prefix + ".0.weight"] = (Hd, DIMS[3])
        s[prefix + ".0.bias"] = (Hd,)
        s[prefix + ".1.weight"] = (Hd,)
        s[prefix + ".1.bias"] = (Hd,)

    def cls_head(prefix, n):
        s[prefix + ".linear_bias"] = (n,)
        s[prefix + ".linear.weight"] = (n, Hd)
        s[prefix + ".linear.bias"] = (n,)

    lt = cfg.loss_type
    if lt["mlm"] == 1:
        embed("mlm_head_embed")
        s["mlm_head.bias"] = (VOCAB,)
        s["mlm_head.transform.dense.weight"] = (Hd, Hd)
        s["mlm_head.transform.dense.bias"] = (Hd,)
        s["mlm_head.transform.LayerNorm.weight"] = (Hd,)
        s["mlm_head.transform.LayerNorm.bias"] = (Hd,)
        s["mlm_head.mlm_decoder.weight"] = (VOCAB, Hd)     # tied to word_embeddings
    if lt["itm"] == 1:
        embed("itm_head_embed")
        cls_head("itm_head", 2)
    if lt["cls"] == 1:
        embed("sup_cls_head_embed")
        cls_head("sup_cls_head", N_SUP)
        embed("sub_cls_head_embed")
        cls_head("sub_cls_head", N_SUB)
    if lt["t2i"] == 1:
        ch = 64
        for name, cin, cout in ITG_CONVS(ch):
            p = f"t2i_head.{name}"
            s[p + ".0.weight"] = (cout, cin, 3, 3)
            s[p + ".1.weight"] = (cout,)
            s[p + ".1.bias"] = (cout,)
            s[p + ".1.running_mean"] = (cout,)
            s[p + ".1.running_var"] = (cout,)
            s[p + ".1.num_batches_tracked"] = ()
        s["t2i_head.score.0.weight"] = (3, 3 * ch, 1, 1)
        s["t2i_head.score.0.bias"] = (3,)
    return s


def ITG_CONVS(ch=64):   # libs/vl_heads.py:116-129, registration order
    return [("reduction1", DIMS[1], ch), ("reduction2", DIMS[2], ch), ("reduction3", DIMS[3], ch),
            ("conv_upsample1", ch, ch), ("conv_upsample2", ch, ch), ("conv_upsample3", ch, ch),
            ("conv_upsample4", ch, ch), ("conv_upsample5", 2 * ch, 2 * ch),
            ("conv_concat2", 2 * ch, 2 * ch), ("conv_concat3", 3 * ch, 3 * ch), ("conv4", 3 * ch, 3 * ch)]


TIED = ("mlm_head.mlm_decoder.weight", "text_embeddings.word_embeddings.weight")


def filled_state_dict(cfg, seed, bert_position_ids=False):
    """torch state_dict from the deterministic filler (tied tensor shared)."""
    from . import filler
    out = OrderedDict()
    for k, shp in param_shapes(cfg, bert_position_ids).items():
        if k == TIED[0]:
            continue
        out[k] = torch.from_numpy(filler.fill_param(seed, k, shp))
    if cfg.loss_type["mlm"] == 1:
        out[TIED[0]] = out[TIED[1]]
    return out


# --------------------------------------------------------------------------- forward
def _ln(x, sd, prefix, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def _lin(x, sd, prefix):
    return F.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"])


def gelu_erf(x):    # libs/vl_heads.py:13-14 and nn.GELU in libs/pvlt.py:56
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def bert_embeddings(sd, ids, keep=None, p=0.1):
    """LN(word[ids] + type[0] + pos[:T]) then dropout.  keep: optional (B,T,H) 0/1 mask
    (train mode with injected randomness); None = eval."""
    T = ids.shape[1]
    # nn.Embedding(..., padding_idx=0): the lookup of PAD rows contributes NO gradient to row 0
    # (row 0 still gets the tied decoder's gradient)
    e = F.embedding(ids, sd["text_embeddings.word_embeddings.weight"], padding_idx=0)
    e = e + sd["text_embeddings.token_type_embeddings.weight"][0]
    e = e + sd["text_embeddings.position_embeddings.weight"][:T]
    e = _ln(e, sd, "text_embeddings.LayerNorm", EPS_BERT)
    if keep is not None:
        e = e * keep / (1.0 - p)
    return e


def pos_embed_for(sd, cfg, stage, H, W):
    """libs/pvlt.py:291-297,341-344 incl. the quirk that every stage compares H*W with
    stage-1's constructor-time patch count."""
    pe = sd[f"pos_embed{stage+1}"]
    if stage == 3:
        pe = pe[:, 1:]
    if H * W == cfg.grid(0) ** 2:
        return pe
    g = cfg.grid(stage)
    pe = pe.reshape(1, g, g, -1).permute(0, 3, 1, 2)
    pe = F.interpolate(pe, size=(H, W), mode="bilinear")
    return pe.reshape(1, -1, H * W).permute(0, 2, 1)


def attention(sd, p, x, H, W, T, heads, sr):
    B, N, C = x.shape
    hd = C // heads
    q = _lin(x, sd, p + "q").reshape(B, N, heads, hd).permute(0, 2, 1, 3)
    if sr > 1:
        img, txt = x[:, : H * W], x[:, H * W:]
        img = img.transpose(1, 2).reshape(B, C, H, W)
        img = F.conv2d(img, sd[p + "sr.weight"], sd[p + "sr.bias"], stride=sr)
        img = _ln(img.reshape(B, C, -1).transpose(1, 2), sd, p + "norm", EPS_DEFAULT)
        src = torch.cat((img, txt), dim=1)
    else:
        src = x
    kv = _lin(src, sd, p + "kv").reshape(B, -1, 2, heads, hd).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    a = torch.softmax((q @ k.transpose(-2, -1)) * (hd ** -0.5), dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, N, C)
    return _lin(o, sd, p + "proj")


def _drop_path(x, keep, rate):
    """timm DropPath: x / (1-p) * mask, one Bernoulli per sample. keep: (B,) 0/1 or None."""
    if keep is None or rate == 0.0:
        return x
    return x / (1.0 - rate) * keep.reshape(-1, *([1] * (x.dim() - 1))).to(x.dtype)


def trunk(sd, cfg, images, ids, masks=None, taps=None):
    """masks (train mode): {'bert': (B,T,H) keep, 'droppath': [(B,) keep per block]}"""
    masks = masks or {}
    B = images.shape[0]
    y = bert_embeddings(sd, ids, masks.get("bert"))
    x = images
    img_feats, text_feats = [], []
    blk = 0
    for i in range(4):
        k = 4 if i == 0 else 2
        x = F.conv2d(x, sd[f"patch_embed{i+1}.proj.weight"], sd[f"patch_embed{i+1}.proj.bias"], stride=k)
        H, W = x.shape[2], x.shape[3]
        x = _ln(x.flatten(2).transpose(1, 2), sd, f"patch_embed{i+1}.norm", EPS_DEFAULT)
        y = _ln(_lin(y, sd, f"text_embed{i+1}.0"), sd, f"text_embed{i+1}.1", EPS_DEFAULT)
        x = torch.cat((x + pos_embed_for(sd, cfg, i, H, W), y + sd[f"text_pos_embed{i+1}"]), dim=1)
        for j in range(cfg.depths[i]):
            p = f"block{i+1}.{j}."
            keep = masks["droppath"][blk] if "droppath" in masks else None
            rate = cfg.dpr[blk]
            a = attention(sd, p + "attn.", _ln(x, sd, p + "norm1", EPS_BLOCK), H, W, cfg.T, HEADS[i], SR[i])
            x = x + _drop_path(a, keep, rate)
            h = gelu_erf(_lin(_ln(x, sd, p + "norm2", EPS_BLOCK), sd, p + "mlp.fc1"))
            x = x + _drop_path(_lin(h, sd, p + "mlp.fc2"), keep2(masks, blk), rate)
            if taps is not None:
                taps[p[:-1]] = x
            blk += 1
        x, y = x[:, : H * W], x[:, H * W:]
        x = x.reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous()
        img_feats.append(x)
        text_feats.append(y)
    return img_feats, text_feats


def keep2(masks, blk):
    """the MLP branch draws its own DropPath sample (second call of the same module)"""
    return masks["droppath2"][blk] if "droppath2" in masks else None


def _conv_bn(sd, p, x, train, bn_out=None):
    x = F.conv2d(x, sd[p + ".0.weight"], None, padding=1)
    if train:
        rm, rv = sd[p + ".1.running_mean"].clone(), sd[p + ".1.running_var"].clone()
        x = F.batch_norm(x, rm, rv, sd[p + ".1.weight"], sd[p + ".1.bias"], True, 0.1, 1e-5)
        if bn_out is not None:
            bn_out[p + ".1.running_mean"], bn_out[p + ".1.running_var"] = rm, rv
        return x
    return F.batch_norm(x, sd[p + ".1.running_mean"], sd[p + ".1.running_var"],
                        sd[p + ".1.weight"], sd[p + ".1.bias"], False, 0.1, 1e-5)


def itg_head(sd, f1, f2, f3, train=False, bn_out=None):
    """MIM decoder, libs/vl_heads.py:136-165."""
    cb = lambda name, t: _conv_bn(sd, "t2i_head." + name, t, train, bn_out)
    up = lambda t, s=2: F.interpolate(t, scale_factor=s, mode="bilinear", align_corners=True)
    low, mid, high = cb("reduction1", f1), cb("reduction2", f2), cb("reduction3", f3)
    a = cb("conv_upsample1", up(high)) * mid
    b = cb("conv_upsample2", up(mid)) * cb("conv_upsample3", up(a)) * low
    c = cb("conv_concat2", torch.cat((a, cb("conv_upsample4", up(high))), 1))
    d = cb("conv_concat3", torch.cat((b, cb("conv_upsample5", up(c))), 1))
    e = cb("conv4", d)
    s = F.conv2d(e, sd["t2i_head.score.0.weight"], sd["t2i_head.score.0.bias"])
    return up(s, 8)


def _embed_ln(sd, prefix, x):
    return _ln(_lin(x, sd, prefix + ".0"), sd, prefix + ".1", EPS_DEFAULT)


def heads(sd, cfg, img_feats, text_feats, train=False, bn_out=None):
    lt = cfg.loss_type
    out = dict(mlm_logits=None, itm_logits=None, sup_cls_logits=None, sub_cls_logits=None, t2i_logits=None)
    last = text_feats[-1]
    if lt["mlm"]:
        h = _embed_ln(sd, "mlm_head_embed", last)
        h = _ln(gelu_erf(_lin(h, sd, "mlm_head.transform.dense")), sd, "mlm_head.transform.LayerNorm", EPS_DEFAULT)
        out["mlm_logits"] = F.linear(h, sd["mlm_head.mlm_decoder.weight"]) + sd["mlm_head.bias"]
    cls_tok = last[:, 0:1, :]
    if lt["itm"]:
        h = _embed_ln(sd, "itm_head_embed", cls_tok)
        out["itm_logits"] = _lin(h, sd, "itm_head.linear") + sd["itm_head.linear_bias"]
    if lt["cls"]:
        for nm in ("sup", "sub"):
            h = _embed_ln(sd, f"{nm}_cls_head_embed", cls_tok)
            out[f"{nm}_cls_logits"] = _lin(h, sd, f"{nm}_cls_head.linear") + sd[f"{nm}_cls_head.linear_bias"]
    if lt["t2i"]:
        out["t2i_logits"] = itg_head(sd, img_feats[1], img_feats[2], img_feats[3], train, bn_out)
    return out


def forward(sd, cfg, images, ids, train=False, masks=None, taps=None, bn_out=None):
    img_feats, text_feats = trunk(sd, cfg, images, ids, masks if train else None, taps)
    out = heads(sd, cfg, img_feats, text_feats, train, bn_out)
    if taps is not None:
        for i in range(4):
            taps[f"img_feat{i+1}"], taps[f"text_feat{i+1}"] = img_feats[i], text_feats[i]
    return out


# --------------------------------------------------------------------------- loss step
MLM_W, ITM_W, T2I_W = 1, 1, 10     # engine_grid_masking.py:23


def masked_positions(mlm_labels):
    """The 'masked-index selection': flat row indices (b*T+t) with label != -1, ascending.
    CrossEntropyLoss(ignore_index=-1) (engine_grid_masking.py:84) averages over exactly these."""
    return torch.nonzero(mlm_labels.reshape(-1) != -1, as_tuple=False).reshape(-1)


def losses(out, batch):
    """engine_grid_masking.py:81-102; the SmoothL1 target is always the CLEAN image."""
    res = {}
    total = 0
    if out["mlm_logits"] is not None:
        res["loss_mlm"] = MLM_W * F.cross_entropy(out["mlm_logits"].reshape(-1, VOCAB),
                                                  batch["mlm_labels"].reshape(-1), ignore_index=-1)
        total = total + res["loss_mlm"]
    if out["itm_logits"] is not None:
        res["loss_itm"] = ITM_W * F.cross_entropy(out["itm_logits"].reshape(-1, 2), batch["itm_labels"].reshape(-1))
        total = total + res["loss_itm"]
    if out["sup_cls_logits"] is not None:
        res["loss_sup_cls"] = F.cross_entropy(out["sup_cls_logits"].reshape(-1, N_SUP), batch["sup_cls_labels"].reshape(-1))
        res["loss_sub_cls"] = F.cross_entropy(out["sub_cls_logits"].reshape(-1, N_SUB), batch["sub_cls_labels"].reshape(-1))
        total = total + res["loss_sup_cls"] + res["loss_sub_cls"]
    if out["t2i_logits"] is not None:
        res["loss_t2i"] = T2I_W * F.smooth_l1_loss(out["t2i_logits"], batch["image"])
        total = total + res["loss_t2i"]
    res["total_loss"] = total
    return res


def step_loss(sd, cfg, batch, step_idx=0, train=True, masks=None, bn_out=None):
    """Forward + loss of one engine iteration: even idx sees the clean image, odd idx the
    grid-masked one when t2i is on (engine_grid_masking.py:72-78; with t2i off the build
    forwards every step -- SURVEY.md App. D #1)."""
    use_masked = (step_idx % 2 == 1) and cfg.loss_type["t2i"] == 1
    img = batch["masked_images"] if use_masked else batch["image"]
    out = forward(sd, cfg, img, batch["input_ids"], train=train, masks=masks, bn_out=bn_out)
    return losses(out, batch), out


def to_torch_batch(np_batch):
    return {k: torch.from_numpy(v) for k, v in np_batch.items()}


# --------------------------------------------------------------------------- optimizer + engine loop
def adamw_param_groups(named_params, weight_decay):
    """timm==0.3.2 optim_factory.add_weight_decay as create_optimizer applies it (call site main_vl.py:308; timm is not
    vendored): 1-D tensors and names ending in '.bias' get weight_decay 0, everything else `weight_decay`; group order
    [no_decay, decay].  named_params: iterable of (name, tensor); a tied tensor must appear once."""
    no_decay, decay = [], []
    for name, p in named_params:
        (no_decay if (p.dim() == 1 or name.endswith(".bias")) else decay).append(p)
    return [dict(params=no_decay, weight_decay=0.0), dict(params=decay, weight_decay=weight_decay)]


def train_loop(sd, cfg, batches, masks_per_iter, lr, weight_decay, betas=(0.9, 0.999), eps=1e-8):
    """The reference's epoch loop on the functional model (engine_grid_masking.py:38-143): per iteration idx -- clean image
    on even idx, grid-masked on odd idx when t2i is on (:72-78), the loss composition (:81-102), optimizer.zero_grad(),
    backward, AdamW step (:122-127; torch.optim.AdamW with timm's param split, main_vl.py:308) -- carrying the BatchNorm
    running statistics from one iteration to the next.  Returns (list of per-iteration loss dicts, final state dict)."""
    leaf = OrderedDict()
    for k, v in sd.items():
        if k == TIED[0]:
            continue
        leaf[k] = v.clone().requires_grad_(True) if (v.is_floating_point() and "running_" not in k) else v.clone()
    if cfg.loss_type["mlm"]:
        leaf[TIED[0]] = leaf[TIED[1]]
    named = [(k, v) for k, v in leaf.items() if k != TIED[0] and v.requires_grad]
    opt = torch.optim.AdamW(adamw_param_groups(named, weight_decay), lr=lr, betas=betas, eps=eps)
    hist = []
    for idx, batch in enumerate(batches):
        bn_out = {}
        ls, _ = step_loss(leaf, cfg, batch, idx, train=True, masks=masks_per_iter[idx] if masks_per_iter else None, bn_out=bn_out)
        opt.zero_grad()
        ls["total_loss"].backward()
        opt.step()
        for k, v in bn_out.items():
            leaf[k] = v.detach()
        for k in list(leaf):
            if k.endswith("num_batches_tracked") and cfg.loss_type["t2i"]:
                leaf[k] = leaf[k] + 1
        hist.append({k: float(v) for k, v in ls.items()})
    return hist, OrderedDict((k, v.detach()) for k, v in leaf.items())
