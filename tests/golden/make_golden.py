#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (/root/reference) on filler
weights and inputs, after checking that oracle/pvlt_oracle.py agrees with it.

Runs only in the build container (needs /root/reference); the fixtures it writes are
committed and travel to the GPU box, the reference does not.

Recipe: the reference model imports on CPU given
  (1) a throw-away stub of the five timm symbols libs/pvlt.py:6-8 uses, created in a
      temp dir at run time (DropPath/trunc_normal_ semantics of timm==0.3.2), and
  (2) BertConfig.from_pretrained patched to the default BertConfig() (== bert-base-uncased
      for every field BertEmbeddings reads; there is no network/HF cache here).
Train-mode randomness (BertEmbeddings dropout, DropPath) is injected as fixed keep-masks
from the filler, in both the reference and the oracle.

usage: python tests/golden/make_golden.py [--add] [case ...]
"""
import os
import sys
import tempfile
import textwrap

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import filler  # noqa: E402
from oracle import pvlt_oracle as O  # noqa: E402

SEED = 20240611

CASES = {
    # name: variant, img, T, B, loss_type, drop_path, train_step?
    "tiny256_pretrain": dict(variant="pvlt_tiny", img=256, T=128, B=4, lt=dict(mlm=1, itm=1, t2i=1, cls=0), dp=0.1, train=True),
    "tiny256_ft": dict(variant="pvlt_tiny", img=256, T=128, B=4, lt=dict(mlm=0, itm=0, t2i=0, cls=1), dp=0.1, train=True),
    "tiny256_all": dict(variant="pvlt_tiny", img=256, T=128, B=2, lt=dict(mlm=1, itm=1, t2i=1, cls=1), dp=0.0, train=False),
    "tiny224_pretrain": dict(variant="pvlt_tiny", img=224, T=128, B=1, lt=dict(mlm=1, itm=1, t2i=1, cls=0), dp=0.0, train=False),
    "tiny384_pretrain": dict(variant="pvlt_tiny", img=384, T=128, B=1, lt=dict(mlm=1, itm=1, t2i=1, cls=0), dp=0.0, train=False),
    "medium384_pretrain": dict(variant="pvlt_medium", img=384, T=128, B=1, lt=dict(mlm=1, itm=1, t2i=1, cls=0), dp=0.0, train=True),
    "small96_T20_ragged": dict(variant="pvlt_small", img=96, T=20, B=3, lt=dict(mlm=1, itm=1, t2i=1, cls=1), dp=0.1, train=True),
    # round 6: the step at sizes that select the large-M kernel variants (store-once attention backward from
    # B x heads >= 512, >= 8-split partial tiles + deferred folds, whole-round 8-phase tiles), config #4 at batch 8, and the fourth factory
    "tiny256_pretrain_b64": dict(variant="pvlt_tiny", img=256, T=128, B=64, lt=dict(mlm=1, itm=1, t2i=1, cls=0), dp=0.1, train=True, steps=(1,)),
    "medium384_pretrain_b8": dict(variant="pvlt_medium", img=384, T=128, B=8, lt=dict(mlm=1, itm=1, t2i=1, cls=0), dp=0.1, train=True, steps=(1,)),
    "large96_T20": dict(variant="pvlt_large", img=96, T=20, B=3, lt=dict(mlm=1, itm=1, t2i=1, cls=1), dp=0.1, train=True, steps=(1,)),
    # BASELINE configuration #5 (CLS fine-tune, scripts_dws/configs/dws_mvlt_ft_exp48.py:11) at a batch that selects the same large-M kernel variants on the CLS-only path
    "tiny256_ft_b64": dict(variant="pvlt_tiny", img=256, T=128, B=64, lt=dict(mlm=0, itm=0, t2i=0, cls=1), dp=0.1, train=True, steps=(0,)),
}


# engine loop (train_one_epoch_vl semantics incl. AdamW): per-iteration losses, parameter deltas and BN statistics after `iters`
# iterations over `iters` different batches (engine_grid_masking.py:38-143, main_vl.py:308)
LOOP_CASES = {
    "tiny256_loop": dict(variant="pvlt_tiny", img=256, T=128, B=4, lt=dict(mlm=1, itm=1, t2i=1, cls=0), dp=0.1, iters=4, lr=1e-3, wd=0.05),
    "tiny256_ft_loop": dict(variant="pvlt_tiny", img=256, T=128, B=4, lt=dict(mlm=0, itm=0, t2i=0, cls=1), dp=0.1, iters=3, lr=1e-3, wd=0.05),
}

# eval callers (engine_grid_masking.py:153-474 with libs/vl_scores.py): per-batch metrics of evaluate_vl, the 101-candidate
# ITM ranking of evaluate_retrieval, the predictions and sklearn metrics of evaluate_recognition
EVAL_CASES = {
    "tiny128_eval": dict(variant="pvlt_tiny", img=128, T=32, B=6, nb=2, lt=dict(mlm=1, itm=1, t2i=1, cls=1), n_query=2, n_cand=101),
}


def install_shims():
    d = tempfile.mkdtemp(prefix="mvlt_shim_")
    os.makedirs(os.path.join(d, "timm", "models"))
    open(os.path.join(d, "timm", "__init__.py"), "w").close()
    open(os.path.join(d, "timm", "models", "__init__.py"), "w").close()
    with open(os.path.join(d, "timm", "models", "layers.py"), "w") as f:
        f.write(textwrap.dedent('''
            import torch, torch.nn as nn
            def to_2tuple(x):
                return tuple(x) if isinstance(x, (tuple, list)) else (x, x)
            def trunc_normal_(t, mean=0., std=1., a=-2., b=2.):
                return nn.init.trunc_normal_(t, mean, std, a, b)
            class DropPath(nn.Module):
                QUEUE = None            # list of (B,) keep masks, consumed in call order
                def __init__(self, drop_prob=None):
                    super().__init__(); self.drop_prob = drop_prob
                def forward(self, x):
                    if self.drop_prob == 0. or not self.training:
                        return x
                    kp = 1 - self.drop_prob
                    shape = (x.shape[0],) + (1,) * (x.ndim - 1)
                    if DropPath.QUEUE is not None:
                        m = DropPath.QUEUE.pop(0).reshape(shape).to(x.dtype)
                    else:
                        m = (kp + torch.rand(shape, dtype=x.dtype, device=x.device)).floor_()
                    return x.div(kp) * m
        '''))
    with open(os.path.join(d, "timm", "models", "registry.py"), "w") as f:
        f.write("def register_model(fn):\n    return fn\n")
    with open(os.path.join(d, "timm", "models", "vision_transformer.py"), "w") as f:
        f.write("def _cfg(url='', **kw):\n    return dict(url=url, **kw)\n")
    sys.path[:0] = [d, "/root/reference"]
    from transformers.models.bert.modeling_bert import BertConfig
    BertConfig.from_pretrained = classmethod(lambda cls, *a, **k: BertConfig())


class FixedDropout(torch.nn.Module):
    def __init__(self, keep, p):
        super().__init__()
        self.keep, self.p = keep, p

    def forward(self, x):
        return x * self.keep / (1.0 - self.p) if self.training else x


def make_masks(cfg, B, T, seed):
    nblk = sum(cfg.depths)
    bert = torch.from_numpy((filler.unit(seed, "bertdrop", B * T * 768) >= 0.1).astype(np.float32)).reshape(B, T, 768)
    d1, d2 = [], []
    for k in range(nblk):
        r = cfg.dpr[k]
        d1.append(torch.from_numpy((filler.unit(seed, f"dp1.{k}", B) >= r).astype(np.float32)))
        d2.append(torch.from_numpy((filler.unit(seed, f"dp2.{k}", B) >= r).astype(np.float32)))
    return dict(bert=bert, droppath=d1, droppath2=d2)


def sample(t, n=64):
    """n values at an ODD stride (so the sample walks through every channel / column, not one of them)."""
    f = t.detach().reshape(-1).to(torch.float32)
    stride = max(1, f.numel() // n) | 1
    return f[::stride][:n].numpy().copy()


def stats(t):
    d = t.detach().to(torch.float64)
    return np.array([d.sum().item(), d.abs().sum().item(), (d * d).sum().sqrt().item()])


def relerr(a, b):
    a, b = a.detach().double(), b.detach().double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def run_case(name, c, add_only=False):
    from libs import pvlt as ref_pvlt
    from timm.models.layers import DropPath
    cfg = O.Cfg(c["variant"], c["lt"], 224, 768, c["T"], c["dp"])
    sd = O.filled_state_dict(cfg, SEED)
    ref = getattr(ref_pvlt, c["variant"])(pretrained=True, token_hidden_size=768, num_text_tokens=c["T"],
                                         loss_type=c["lt"], pretrained_pth=None, drop_path_rate=c["dp"],
                                         drop_rate=0.0, num_classes=1000, in_chans=3)
    ref_keys = list(ref.state_dict().keys())
    want = [k for k in O.param_shapes(cfg).keys()]
    assert ref_keys == want, ("state_dict schema/order mismatch", [k for k in ref_keys if k not in want], [k for k in want if k not in ref_keys])
    missing = ref.load_state_dict(sd, strict=True)
    assert ref.mlm_head.mlm_decoder.weight is ref.text_embeddings.word_embeddings.weight if c["lt"]["mlm"] else True
    nb = filler.make_batch(SEED, c["B"], c["img"], c["T"])
    batch = O.to_torch_batch(nb)
    G = {}
    G["meta"] = np.array([SEED, c["B"], c["img"], c["T"], c["dp"]], dtype=np.float64)

    # ---------------- eval forward
    ref.eval()
    taps_ref = {}
    hooks = []
    for i in range(4):
        for j, blk in enumerate(getattr(ref, f"block{i+1}")):
            hooks.append(blk.register_forward_hook(lambda m, a, o, key=f"block{i+1}.{j}": taps_ref.__setitem__(key, o)))
    with torch.no_grad():
        img_feats, text_feats = ref.forward_pyramid_features_vl(batch["image"], batch["input_ids"])
        out_ref = ref(batch["image"], batch["input_ids"])
    for h in hooks:
        h.remove()
    for i in range(4):
        taps_ref[f"img_feat{i+1}"], taps_ref[f"text_feat{i+1}"] = img_feats[i], text_feats[i]
    taps = {}
    with torch.no_grad():
        out = O.forward(sd, cfg, batch["image"], batch["input_ids"], train=False, taps=taps)
    worst = 0.0
    for k, v in taps_ref.items():
        e = relerr(taps[k], v)
        if e > 5e-6:
            print("   note: tap", k, "rel err", e)
        worst = max(worst, e)
        G[f"eval/tap/{k}/stats"] = stats(v)
        G[f"eval/tap/{k}/sample"] = sample(v, 1024)
    for k, v in out_ref.items():
        if v is None:
            assert out[k] is None
            continue
        e = relerr(out[k], v)
        if e > 5e-6:
            print("   note: out", k, "rel err", e)
        worst = max(worst, e)
        G[f"eval/out/{k}/stats"] = stats(v)
        G[f"eval/out/{k}/sample"] = sample(v, 4096)
    assert worst < 5e-5, (name, "oracle != reference (eval)", worst)
    for k in ("itm_logits", "sup_cls_logits", "sub_cls_logits"):
        if out_ref[k] is not None:
            G[f"eval/full/{k}"] = out_ref[k].numpy().copy()
    # noise floor of bf16 itself: the REFERENCE under torch.autocast(bf16) against its own fp32 forward (relative L2)
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        out_bf = ref(batch["image"], batch["input_ids"])
    for k, v in out_ref.items():
        if v is not None:
            G[f"eval/bf16_floor/{k}"] = np.array(relerr(out_bf[k].float(), v))
    print(f"[{name}] reference bf16-autocast vs its fp32 (rel L2):", {k: round(float(G[f'eval/bf16_floor/{k}']), 4) for k, v in out_ref.items() if v is not None})
    pos = O.masked_positions(batch["mlm_labels"])
    G["masked_positions"] = pos.numpy().astype(np.int64)
    if out_ref["mlm_logits"] is not None:
        rows = out_ref["mlm_logits"].reshape(-1, O.VOCAB)[pos]
        tv, ti = rows.topk(8, dim=-1)
        G["eval/mlm/top8_val"] = tv.numpy().copy()
        G["eval/mlm/top8_idx"] = ti.numpy().astype(np.int64)
        G["eval/mlm/label_logit"] = rows.gather(1, batch["mlm_labels"].reshape(-1)[pos][:, None]).numpy().copy()
    if out_ref["t2i_logits"] is not None:
        t = out_ref["t2i_logits"]
        s = max(1, c["img"] // 16)
        bs = max(1, c["B"] // 8)                 # large batches: every bs-th image (the fixtures stay small)
        G["eval/t2i/grid"] = t[::bs, :, ::s, ::s].numpy().copy()
        if bs > 1:
            G["eval/t2i/grid_bstride"] = np.array(bs)
    l_eval = O.losses(out_ref, batch)
    for k, v in l_eval.items():
        G[f"eval/loss/{k}"] = np.array(float(v))
    print(f"[{name}] eval oracle-vs-reference worst rel err {worst:.2e}; losses", {k: round(float(v), 5) for k, v in l_eval.items()})

    # ---------------- train-mode step (forward + loss + backward), injected masks
    if c["train"]:
        for step_idx in c.get("steps", (0, 1)):
            if step_idx == 1 and not c["lt"]["t2i"]:
                continue
            masks = make_masks(cfg, c["B"], c["T"], SEED + step_idx)
            ref.train()
            ref.load_state_dict(sd, strict=True)        # reset BN running stats
            ref.text_embeddings.dropout = FixedDropout(masks["bert"], 0.1)
            q = []
            for k in range(sum(cfg.depths)):
                if cfg.dpr[k] > 0:
                    q += [masks["droppath"][k], masks["droppath2"][k]]
            DropPath.QUEUE = q
            ref.zero_grad()
            img = batch["masked_images"] if (step_idx % 2 == 1) else batch["image"]
            # how far the per-pair terms of itm_head_embed.0.bias's gradient cancel in the batch sum: c = sqrt(B sum_b |g_b|^2) / |sum_b g_b| >= 1 (the
            # bf16 gate of tests/test_model_gpu.py prices that tensor against the un-cancelled scale, as it does the three ITM biases behind it)
            caught, hk = {}, None
            if c["lt"]["itm"]:
                hk = ref.itm_head_embed[0].register_full_backward_hook(lambda m, gi, go: caught.__setitem__("g", go[0].detach().double()))
            out_r = ref(img, batch["input_ids"])
            l_r = O.losses(out_r, batch)
            l_r["total_loss"].backward()
            if hk is not None:
                hk.remove()
                gb = caught["g"].reshape(c["B"], -1)
                G[f"train{step_idx}/cancel/itm_head_embed.0.bias"] = np.array(float(np.sqrt(c["B"] * (gb ** 2).sum().item()) / max(1e-30, gb.sum(0).norm().item())))
            assert len(q) == 0
            DropPath.QUEUE = None
            g_ref = {k: p.grad for k, p in ref.named_parameters()}

            sdg = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and "running_" not in k) else v)
                   for k, v in sd.items() if k != O.TIED[0]}
            if c["lt"]["mlm"]:
                sdg[O.TIED[0]] = sdg[O.TIED[1]]
            bn_out = {}
            l_o, out_o = O.step_loss(sdg, cfg, batch, step_idx, train=True, masks=masks, bn_out=bn_out)
            l_o["total_loss"].backward()
            worst = 0.0
            for k, v in l_r.items():
                worst = max(worst, abs(float(v) - float(l_o[k])) / max(1e-12, abs(float(v))))
                G[f"train{step_idx}/loss/{k}"] = np.array(float(v))
            gn = {}
            for k, g in g_ref.items():
                if g is None:
                    continue
                e = relerr(sdg[k].grad, g)
                worst = max(worst, e)
                gn[k] = g.double().norm().item()
                G[f"train{step_idx}/grad/{k}/norm"] = np.array(gn[k])
                G[f"train{step_idx}/grad/{k}/sample"] = sample(g, 32)
            bn_ref = {k: v for k, v in ref.state_dict().items() if "running_" in k}
            for k, v in bn_ref.items():
                worst = max(worst, relerr(bn_out[k], v))
                G[f"train{step_idx}/bn/{k}"] = v.numpy().copy()
            assert worst < 2e-4, (name, "oracle != reference (train)", worst)
            print(f"[{name}] train step {step_idx} oracle-vs-reference worst rel err {worst:.2e}; losses",
                  {k: round(float(v), 5) for k, v in l_r.items()})
            ref.text_embeddings.dropout = torch.nn.Dropout(0.1)

    path = os.path.join(HERE, name + ".npz")
    if add_only:
        # keep the committed arrays byte for byte, add the new keys only -- after checking that this run reproduces the committed ones
        old = dict(np.load(path))
        for k, v in old.items():
            assert k in G and np.allclose(np.asarray(G[k], dtype=np.float64), np.asarray(v, dtype=np.float64), rtol=1e-4, atol=1e-6), (name, "this run does not reproduce the committed fixture", k)
        added = [k for k in G if k not in old]
        G = {**old, **{k: G[k] for k in added}}
        print(f"[{name}] --add: {len(old)} committed arrays reproduced and kept, added {added}")
    np.savez_compressed(path, **G)
    print(f"[{name}] wrote {path} ({os.path.getsize(path)/1024:.1f} KiB, {len(G)} arrays)")


def loop_view(name, delta):
    """The part of a parameter delta that is comparable between implementations after AdamW steps.  The K half of every
    `attn.kv.bias` has a mathematically ZERO gradient (softmax is invariant to a per-query shift of the scores, and a key
    bias shifts all scores of a query by q.b), so what reaches AdamW there is pure rounding noise, which m / sqrt(v)
    normalises to full-size +-lr steps: the reference and the oracle -- both fp32 PyTorch -- already disagree by 8-16 % on
    that half.  Only the V half is compared."""
    if name.endswith("attn.kv.bias"):
        return delta[delta.numel() // 2:]
    return delta


def build_ref(c, T, dp):
    from libs import pvlt as ref_pvlt
    cfg = O.Cfg(c["variant"], c["lt"], 224, 768, T, dp)
    sd = O.filled_state_dict(cfg, SEED)
    ref = getattr(ref_pvlt, c["variant"])(pretrained=True, token_hidden_size=768, num_text_tokens=T, loss_type=c["lt"],
                                         pretrained_pth=None, drop_path_rate=dp, drop_rate=0.0, num_classes=1000, in_chans=3)
    ref.load_state_dict(sd, strict=True)
    return cfg, sd, ref


def run_loop_case(name, c):
    """The reference model driven the way train_one_epoch_vl drives it (engine_grid_masking.py:38-143; the engine module itself
    needs timm.data / torchvision / mmcv and a GPU, so its loop body is restated here around the REAL model and
    torch.optim.AdamW with timm's parameter split), checked against oracle.train_loop."""
    from timm.models.layers import DropPath
    cfg, sd, ref = build_ref(c, c["T"], c["dp"])
    ref.train()
    named = [(k, p) for k, p in ref.named_parameters()]           # named_parameters() de-duplicates the tied decoder weight
    opt = torch.optim.AdamW(O.adamw_param_groups(named, c["wd"]), lr=c["lr"], betas=(0.9, 0.999), eps=1e-8)
    batches = [O.to_torch_batch(filler.make_batch(SEED + 100 * it, c["B"], c["img"], c["T"])) for it in range(c["iters"])]
    masks = [make_masks(cfg, c["B"], c["T"], SEED + it) for it in range(c["iters"])]
    G = {"meta": np.array([SEED, c["B"], c["img"], c["T"], c["dp"], c["iters"], c["lr"], c["wd"]], dtype=np.float64)}
    hist_ref = []
    for idx, batch in enumerate(batches):
        ref.text_embeddings.dropout = FixedDropout(masks[idx]["bert"], 0.1)
        q = []
        for k in range(sum(cfg.depths)):
            if cfg.dpr[k] > 0:
                q += [masks[idx]["droppath"][k], masks[idx]["droppath2"][k]]
        DropPath.QUEUE = q
        img = batch["masked_images"] if (idx % 2 == 1 and c["lt"]["t2i"]) else batch["image"]
        out = ref(img, batch["input_ids"])
        ls = O.losses(out, batch)
        opt.zero_grad()
        ls["total_loss"].backward()
        opt.step()
        assert len(q) == 0
        DropPath.QUEUE = None
        hist_ref.append({k: float(v) for k, v in ls.items()})
    hist_o, sd_o = O.train_loop(sd, cfg, batches, masks, c["lr"], c["wd"])
    worst = 0.0
    for it, (hr, ho) in enumerate(zip(hist_ref, hist_o)):
        for k, v in hr.items():
            worst = max(worst, abs(v - ho[k]) / max(1e-12, abs(v)))
            G[f"loop/loss/{it}/{k}"] = np.array(v)
    sd_ref = ref.state_dict()
    worst_d = 0.0
    for k, v in sd_ref.items():
        if k == O.TIED[0]:
            continue
        if not v.is_floating_point():
            assert int(v) == int(sd_o[k]) == c["iters"], (k, int(v), int(sd_o[k]))
            G[f"loop/int/{k}"] = np.array(int(v))
            continue
        d_ref = loop_view(k, (v - sd[k]).double())
        d_o = loop_view(k, (sd_o[k] - sd[k]).double())
        if d_ref.norm().item() > 0:
            e = ((d_o - d_ref).norm() / d_ref.norm()).item()
            if e > 1e-3:
                print("   note: delta", k, "rel err", round(e, 5))
            worst_d = max(worst_d, e)
        G[f"loop/delta/{k}/norm"] = np.array(d_ref.norm().item())
        G[f"loop/delta/{k}/sample"] = sample(d_ref, 64)
    assert worst < 2e-4 and worst_d < 5e-3, (name, "oracle != reference (loop)", worst, worst_d)
    print(f"[{name}] {c['iters']} engine iterations: oracle-vs-reference worst loss rel err {worst:.2e}, worst parameter-delta rel err {worst_d:.2e}")
    print(f"[{name}] losses per iteration:", [round(h['total_loss'], 5) for h in hist_ref])
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **G)
    print(f"[{name}] wrote {path} ({os.path.getsize(path)/1024:.1f} KiB, {len(G)} arrays)")


def eval_batch(seed, it, B, img, T, ref):
    """a filler batch whose labels are made half right on purpose: for every second masked position / sample the label is the
    reference model's own top-1, so the accuracies land near 0.5 and react to any argmax that moves"""
    nb = filler.make_batch(seed + 100 * it, B, img, T)
    batch = O.to_torch_batch(nb)
    with torch.no_grad():
        o_mlm = ref(batch["image"], batch["input_ids"])
        o_1 = ref(batch["image"], batch["ori_input_ids"])
    pos = O.masked_positions(batch["mlm_labels"])
    lab = batch["mlm_labels"].reshape(-1).clone()
    top = o_mlm["mlm_logits"].reshape(-1, O.VOCAB)[pos].argmax(-1)
    lab[pos[::2]] = top[::2]
    batch["mlm_labels"] = lab.reshape(B, T)
    for key, lk in (("itm_logits", "itm_labels"), ("sup_cls_logits", "sup_cls_labels"), ("sub_cls_logits", "sub_cls_labels")):
        top = o_1[key].reshape(B, -1).argmax(-1)
        n = o_1[key].shape[-1]
        l = batch[lk].reshape(-1).clone()
        l[::2] = top[::2]
        l[1::2] = (top[1::2] + 1) % n
        batch[lk] = l.reshape(B, 1)
    return batch


def run_eval_case(name, c):
    from libs import vl_scores as VS                      # the reference's own metric functions (torch + math only)
    from sklearn.metrics import accuracy_score, f1_score
    cfg, sd, ref = build_ref(c, c["T"], 0.0)
    ref.eval()
    B, img, T = c["B"], c["img"], c["T"]
    G = {"meta": np.array([SEED, B, img, T, c["nb"], c["n_query"], c["n_cand"]], dtype=np.float64)}
    # ---- evaluate_vl (engine_grid_masking.py:153-333), per batch
    keys = ("mlm_acc", "itm_acc", "sup_cls_acc", "sub_cls_acc", "t2i_psnr", "total_loss")
    sums = {k: 0.0 for k in keys}
    for it in range(c["nb"]):
        batch = eval_batch(SEED, it, B, img, T, ref)
        for k in ("mlm_labels", "itm_labels", "sup_cls_labels", "sub_cls_labels"):
            G[f"vl/{it}/{k}"] = batch[k].numpy().copy()
        with torch.no_grad():
            o_mlm = ref(batch["image"], batch["input_ids"])
            o_1 = ref(batch["image"], batch["ori_input_ids"])
            o_3 = ref(batch["masked_images"], batch["ori_input_ids"])
        total = 0.0
        m = {}
        total += 1 * torch.nn.CrossEntropyLoss(ignore_index=-1)(o_mlm["mlm_logits"].view(-1, 30522), batch["mlm_labels"].view(-1)).item()
        m["mlm_acc"] = VS.compute_mlm_score(o_mlm["mlm_logits"], batch["mlm_labels"])
        total += 1 * torch.nn.CrossEntropyLoss()(o_1["itm_logits"].view(-1, 2), batch["itm_labels"].view(-1)).item()
        m["itm_acc"] = VS.compute_score_with_logits(o_1["itm_logits"].view(-1, 2), batch["itm_labels"].view(-1)).sum().item() / B
        total += torch.nn.CrossEntropyLoss()(o_1["sup_cls_logits"].view(-1, 48), batch["sup_cls_labels"].view(-1)).item()
        total += torch.nn.CrossEntropyLoss()(o_1["sub_cls_logits"].view(-1, 122), batch["sub_cls_labels"].view(-1)).item()
        m["sup_cls_acc"] = VS.compute_score_with_logits(o_1["sup_cls_logits"].view(-1, 48), batch["sup_cls_labels"].view(-1)).sum().item() / B
        m["sub_cls_acc"] = VS.compute_score_with_logits(o_1["sub_cls_logits"].view(-1, 122), batch["sub_cls_labels"].view(-1)).sum().item() / B
        total += 10 * torch.nn.SmoothL1Loss()(o_3["t2i_logits"], batch["image"]).item()
        m["t2i_psnr"] = VS.compute_psnr(o_3["t2i_logits"], batch["image"])
        m["total_loss"] = total
        for k in keys:
            G[f"vl/{it}/{k}"] = np.array(float(m[k]))
            sums[k] += float(m[k]) * B
        # recognition predictions of the same forward (evaluate_recognition, :396-474)
        G[f"recog/{it}/sup_pred"] = torch.max(torch.softmax(o_1["sup_cls_logits"].view(-1, 48), -1), -1)[1].numpy().copy()
        G[f"recog/{it}/sub_pred"] = torch.max(torch.softmax(o_1["sub_cls_logits"].view(-1, 122), -1), -1)[1].numpy().copy()
        G[f"recog/{it}/sup_top2"] = o_1["sup_cls_logits"].view(-1, 48).topk(2, -1)[0].numpy().copy()
        G[f"recog/{it}/sub_top2"] = o_1["sub_cls_logits"].view(-1, 122).topk(2, -1)[0].numpy().copy()
    for k in keys:
        G[f"vl/avg/{k}"] = np.array(sums[k] / (B * c["nb"]))
    print(f"[{name}] evaluate_vl averages:", {k: round(float(G[f'vl/avg/{k}']), 5) for k in keys})
    sl = np.concatenate([G[f"vl/{it}/sup_cls_labels"].reshape(-1) for it in range(c["nb"])])
    sp = np.concatenate([G[f"recog/{it}/sup_pred"] for it in range(c["nb"])])
    bl = np.concatenate([G[f"vl/{it}/sub_cls_labels"].reshape(-1) for it in range(c["nb"])])
    bp = np.concatenate([G[f"recog/{it}/sub_pred"] for it in range(c["nb"])])
    for tag, l, p_ in (("sup", sl, sp), ("sub", bl, bp)):       # calculate_cls_metrics (:477-486)
        G[f"recog/{tag}_metrics"] = np.array([accuracy_score(l, p_), f1_score(l, p_, average="macro"), f1_score(l, p_, average="micro"),
                                              f1_score(l, p_, average="weighted")])
    print(f"[{name}] recognition metrics (acc, macro, micro, weighted): sup {G['recog/sup_metrics'].round(4)}, sub {G['recog/sub_metrics'].round(4)}")
    # ---- evaluate_retrieval (:336-393): n_cand candidates per query, candidate 0 is the match
    for qi in range(c["n_query"]):
        cand = retrieval_query(SEED, qi, c["n_cand"], img, T)
        with torch.no_grad():
            logits = ref(cand["images_101"].squeeze(), cand["ori_input_ids_101"].squeeze())["itm_logits"].view(-1, 2)
        sm = torch.softmax(logits, dim=-1)
        srt, ind = torch.sort(sm[:, 1], dim=-1, descending=True)
        rank0 = int(np.argwhere(ind.numpy() == 0))
        G[f"retr/{qi}/score"] = sm[:, 1].numpy().copy()
        G[f"retr/{qi}/order"] = ind.numpy().astype(np.int64)
        G[f"retr/{qi}/rank0"] = np.array(rank0)
        print(f"[{name}] retrieval query {qi}: rank of candidate 0 = {rank0}, score spread {float(srt[0]):.4f}..{float(srt[-1]):.4f}")
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **G)
    print(f"[{name}] wrote {path} ({os.path.getsize(path)/1024:.1f} KiB, {len(G)} arrays)")


def retrieval_query(seed, qi, n_cand, img, T):
    """one item of the reference's retrieval loader (mcloader/fashion_gen.py:499-505): `images_101` (1, n, 3, S, S) and
    `ori_input_ids_101` (1, n, T) -- text retrieval style: the image repeated, n captions"""
    nb = filler.make_batch(seed + 1000 + qi, n_cand, img, T)
    images = np.repeat(nb["image"][:1], n_cand, axis=0)
    return dict(images_101=torch.from_numpy(images)[None], ori_input_ids_101=torch.from_numpy(nb["ori_input_ids"])[None],
                info_list=[dict(img_name=f"q{qi}_{j}") for j in range(n_cand)])


def run_batchprep_case(name="batchprep_ref"):
    """Pin the decision logic of oracle/batchprep_oracle.py against the reference's OWN dataset functions
    (mcloader/fashion_gen.py:225-254 generate_grid_mask, :383-409 random_masking_features), run here with their random draws
    replaced by the oracle's Philox-derived ones: np.random.shuffle applies the oracle's permutations, random.random /
    random.choice return the oracle's draws.  The module imports cv2 and torchvision at the top (absent in this image, unused
    by the two functions): empty stand-in modules are registered for the import only."""
    import random
    import types
    from oracle import batchprep_oracle as BP
    for mod in ("cv2", "torchvision", "torchvision.transforms"):
        sys.modules.setdefault(mod, types.ModuleType(mod))
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    from mcloader import fashion_gen as FG
    G = {}
    seed = SEED
    # ---- grid mask: (S, ratio) cases x samples; the reference is driven with the oracle's permutations
    cases = [(256, 0.5), (384, 0.5), (256, 0.75), (224, 0.25)]
    G["grid/cases"] = np.array(cases, dtype=np.float64)
    real_shuffle = np.random.shuffle
    for ci, (S, ratio) in enumerate(cases):
        g = S // 16
        P = g * g
        for sample in range(4):
            k1 = BP.draws(seed, sample, np.arange(P), BP.STREAM_GRID)[0]
            k2 = BP.draws(seed, sample, np.arange(P), BP.STREAM_ROW)[0].reshape(g, g)
            pg, prow = BP.perm_from_keys(k1), [BP.perm_from_keys(k2[i]) for i in range(g)]
            queue = [pg] + prow

            def fake_shuffle(x, queue=queue):
                perm = queue.pop(0)
                assert len(perm) == len(x)
                x[:] = [x[j] for j in perm]

            np.random.shuffle = fake_shuffle
            try:
                m = FG.FashionGenDatasetPreTrain.generate_grid_mask(None, input_size=(S, S), mask_ratio=ratio, patch_size=16)
            finally:
                np.random.shuffle = real_shuffle
            assert not queue and m.shape == (1, S, S)
            ref_flags = m[0, ::16, ::16].astype(np.uint8)
            assert np.array_equal(np.repeat(np.repeat(ref_flags, 16, 0), 16, 1), m[0].astype(np.uint8))      # patches are constant
            mine = BP.grid_flags(seed, sample, g, g, int(ratio * P), 1)
            assert np.array_equal(mine, ref_flags), ("grid mask restatement != reference", S, ratio, sample)
            G[f"grid/{ci}/{sample}"] = ref_flags
        print(f"[{name}] generate_grid_mask S={S} ratio={ratio}: oracle == reference on 4 samples; realised ratios",
              [round(float(G[f'grid/{ci}/{s_}'].mean()), 3) for s_ in range(4)])
    # masked_fill (fashion_gen.py:176) on one sample
    img = filler.unit(seed, "bp_image", 3 * 256 * 256).reshape(3, 256, 256).astype(np.float32)
    flags = G["grid/0/0"]
    mask_full = np.repeat(np.repeat(flags, 16, 0), 16, 1)[None].astype(np.float64)          # (1, S, S) like the reference's img_mask
    ref_masked = torch.from_numpy(img).clone().masked_fill_(torch.Tensor(mask_full).byte().bool(), value=torch.tensor(1e-6)).numpy()
    assert np.array_equal(BP.apply_grid_mask(img, flags), ref_masked)
    G["fill/sample"] = ref_masked[:, ::7, ::5].copy()
    # ---- token masking: the reference method on token strings "<id>", vocab maps them back
    vocab = {str(i): i for i in range(BP.VOCAB)}
    vocab["[MASK]"] = BP.TOK_MASK
    vocab.pop(str(BP.TOK_MASK))
    items = list(vocab.items())
    fake_self = types.SimpleNamespace(word_mask_rate=0.15, tokenizer=types.SimpleNamespace(vocab=vocab))
    T = 128
    real_random, real_choice = random.random, random.choice
    n_sel = 0
    for sample in range(6):
        nb = filler.make_batch(seed + sample, 1, 32, T)
        ori = nb["ori_input_ids"][0]
        L = int(np.nonzero(ori == BP.TOK_SEP)[0][0]) - 1
        x0, x1, x2, _ = BP.draws(seed, sample, np.arange(T), BP.STREAM_TOKEN)
        r1, r2, r3 = x0 >> np.uint32(8), x1 >> np.uint32(8), x2
        state = dict(t=1)

        def fake_random():
            t = state["t"]
            state["t"] += 1
            state["cur"] = t
            # a draw that makes the reference take the oracle's branch: prob < 0.15 iff selected, prob / 0.15 = r2 / 2^24
            return 0.15 * (float(r2[t]) / 2 ** 24) if r1[t] < BP.T15 else 0.5

        def fake_choice(seq):
            return seq[int((int(r3[state["cur"]]) * BP.VOCAB) >> 32)]

        tokens = [str(int(v)) if int(v) != BP.TOK_MASK else "[MASK]" for v in ori[1:1 + L]]
        random.random, random.choice = fake_random, fake_choice
        try:
            out_tokens, lm_label = FG.FashionGenDatasetPreTrain.random_masking_features(fake_self, list(tokens))
        finally:
            random.random, random.choice = real_random, real_choice
        # label / id assembly of text_process (fashion_gen.py:340-364)
        ref_ids = np.zeros(T, dtype=np.int64)
        ref_ids[0], ref_ids[1 + L] = BP.TOK_CLS, BP.TOK_SEP
        ref_ids[1:1 + L] = [vocab[t_] for t_ in out_tokens]
        ref_lab = np.array([-1] + lm_label + [-1] * (T - L - 1) , dtype=np.int64)[:T]
        ids, lab = BP.mask_tokens(seed, sample, ori)
        # items order: the uniform pick indexes list(vocab.items()); with "103" renamed to "[MASK]" and moved to the end the index of
        # id i is i for i < 103 and i - 1 above -- map the oracle's uniform id through the same table for the comparison
        pick = {i: items[i][1] for i in range(BP.VOCAB)}
        ids_cmp = ids.copy()
        rnd_pos = (lab != -1) & (ids != BP.TOK_MASK) & (ids != ori)
        ids_cmp[rnd_pos] = [pick[int(v)] for v in ids[rnd_pos]]
        assert np.array_equal(lab, ref_lab), ("token labels restatement != reference", sample)
        assert np.array_equal(ids_cmp, ref_ids), ("token ids restatement != reference", sample)
        n_sel += int((lab != -1).sum())
        G[f"tok/{sample}/ori"], G[f"tok/{sample}/ids"], G[f"tok/{sample}/labels"] = ori, ids, lab
    print(f"[{name}] random_masking_features: oracle == reference on 6 captions ({n_sel} selected positions)")
    G["meta"] = np.array([seed], dtype=np.float64)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **G)
    print(f"[{name}] wrote {path} ({os.path.getsize(path)/1024:.1f} KiB, {len(G)} arrays)")


def main():
    install_shims()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    add_only = "--add" in sys.argv[1:]          # CASES only: keep the committed arrays, add the keys a newer generator knows
    names = [a for a in sys.argv[1:] if a != "--add"] or (list(CASES) + list(LOOP_CASES) + list(EVAL_CASES) + ["batchprep_ref"])
    for n in names:
        if n == "batchprep_ref":
            run_batchprep_case(n)
        elif n in LOOP_CASES:
            run_loop_case(n, LOOP_CASES[n])
        elif n in EVAL_CASES:
            run_eval_case(n, EVAL_CASES[n])
        else:
            run_case(n, CASES[n], add_only)


if __name__ == "__main__":
    main()
