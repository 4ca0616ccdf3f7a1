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

usage: python tests/golden/make_golden.py [case ...]
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
    "medium384_pretrain": dict(variant="pvlt_medium", img=384, T=128, B=1, lt=dict(mlm=1, itm=1, t2i=1, cls=0), dp=0.0, train=False),
    "small96_T20_ragged": dict(variant="pvlt_small", img=96, T=20, B=3, lt=dict(mlm=1, itm=1, t2i=1, cls=1), dp=0.1, train=True),
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


def run_case(name, c):
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
        G["eval/t2i/grid"] = t[:, :, ::s, ::s].numpy().copy()
    l_eval = O.losses(out_ref, batch)
    for k, v in l_eval.items():
        G[f"eval/loss/{k}"] = np.array(float(v))
    print(f"[{name}] eval oracle-vs-reference worst rel err {worst:.2e}; losses", {k: round(float(v), 5) for k, v in l_eval.items()})

    # ---------------- train-mode step (forward + loss + backward), injected masks
    if c["train"]:
        for step_idx in (0, 1):
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
            out_r = ref(img, batch["input_ids"])
            l_r = O.losses(out_r, batch)
            l_r["total_loss"].backward()
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
    np.savez_compressed(path, **G)
    print(f"[{name}] wrote {path} ({os.path.getsize(path)/1024:.1f} KiB, {len(G)} arrays)")


def main():
    install_shims()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    names = sys.argv[1:] or list(CASES)
    for n in names:
        run_case(n, CASES[n])


if __name__ == "__main__":
    main()
