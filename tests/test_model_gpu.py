"""GPU: the HIP model (mvlt_amd) against the CPU oracle computed live on the same filler weights/inputs AND against
the committed golden vectors captured from the real reference (tests/golden/*.npz).

Tolerances (north_star): fp32 compute path 1e-3, bf16 compute path 2e-2, both as max-abs error normalised by the
reference tensor's max-abs; the masked-index selection must be bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import filler
from oracle import pvlt_oracle as O

pytestmark = pytest.mark.gpu
TOL = {torch.float32: 1e-3, torch.bfloat16: 2e-2}

CASES = {
    "tiny256_pretrain": dict(variant="pvlt_tiny", lt=dict(mlm=1, itm=1, t2i=1, cls=0)),
    "tiny256_ft": dict(variant="pvlt_tiny", lt=dict(mlm=0, itm=0, t2i=0, cls=1)),
    "tiny256_all": dict(variant="pvlt_tiny", lt=dict(mlm=1, itm=1, t2i=1, cls=1)),
    "tiny224_pretrain": dict(variant="pvlt_tiny", lt=dict(mlm=1, itm=1, t2i=1, cls=0)),
    "tiny384_pretrain": dict(variant="pvlt_tiny", lt=dict(mlm=1, itm=1, t2i=1, cls=0)),
    "medium384_pretrain": dict(variant="pvlt_medium", lt=dict(mlm=1, itm=1, t2i=1, cls=0)),
    "small96_T20_ragged": dict(variant="pvlt_small", lt=dict(mlm=1, itm=1, t2i=1, cls=1)),
}


def maxrel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def sample(t, n=64):
    f = t.detach().reshape(-1).to(torch.float32).cpu()
    stride = max(1, f.numel() // n)
    return f[::stride][:n].numpy()


def build(name, golden_dir, dtype):
    from mvlt_amd import pvlt
    c = CASES[name]
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    seed, B, img, T, dp = int(g["meta"][0]), int(g["meta"][1]), int(g["meta"][2]), int(g["meta"][3]), float(g["meta"][4])
    cfg = O.Cfg(c["variant"], c["lt"], 224, 768, T, dp)
    sd = O.filled_state_dict(cfg, seed)
    model = getattr(pvlt, c["variant"])(pretrained=True, token_hidden_size=768, num_text_tokens=T, loss_type=c["lt"],
                                        pretrained_pth=None, drop_path_rate=dp, drop_rate=0.0, num_classes=1000, in_chans=3,
                                        compute_dtype=dtype)
    model.load_state_dict(sd, strict=True)
    model.cuda()
    batch = O.to_torch_batch(filler.make_batch(seed, B, img, T))
    return model, cfg, sd, batch, g, seed


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", list(CASES))
def test_eval_forward_parity(golden_dir, name, dtype):
    model, cfg, sd, batch, g, seed = build(name, golden_dir, dtype)
    model.eval()
    model._taps = {}
    dev = torch.device("cuda:0")
    with torch.no_grad():
        out = model(batch["image"].to(dev), batch["input_ids"].to(dev))
    torch.cuda.synchronize()
    taps_o = {}
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    with torch.no_grad():
        ref = O.forward(sd, cfg, batch["image"], batch["input_ids"], taps=taps_o)
    tol = TOL[dtype]
    errs = {}
    for i in range(4):
        for k in (f"img_feat{i+1}", f"text_feat{i+1}"):
            errs[k] = maxrel(model._taps[k], taps_o[k])
    for k, v in ref.items():
        if v is None:
            assert out[k] is None, k
        else:
            assert tuple(out[k].shape) == tuple(v.shape), (k, out[k].shape, v.shape)
            errs[k] = maxrel(out[k].float(), v)
    bad = {k: e for k, e in errs.items() if not e < tol}
    assert not bad, (name, dtype, bad, errs)
    # against the reference's own numbers (golden fixture)
    for k in g.files:
        if k.startswith("eval/out/") and k.endswith("/sample"):
            key = k.split("/")[2]
            a, b = sample(out[key].float(), 256), g[k]
            assert np.abs(a - b).max() / max(1e-6, np.abs(b).max()) < 2 * tol, k
        if k.startswith("eval/full/"):
            key = k.split("/")[2]
            a, b = out[key].float().cpu().numpy(), g[k]
            assert np.abs(a - b).max() / max(1e-6, np.abs(b).max()) < 2 * tol, k
    # masked-index selection, bit-exact
    from mvlt_amd import ops
    lab = batch["mlm_labels"].to(dev).reshape(-1).contiguous()
    idx = torch.empty(lab.numel(), device=dev, dtype=torch.int32)
    cnt = torch.zeros(1, device=dev, dtype=torch.int32)
    ops.masked_select(lab, idx, cnt)
    n = int(cnt.item())
    assert np.array_equal(idx[:n].cpu().numpy().astype(np.int64), g["masked_positions"])


def _losses_like_engine(out, batch, dev):
    """same composition as reference engine_grid_masking.py:81-102 (fused MLM loss when present)."""
    import torch.nn.functional as F
    total = 0
    res = {}
    if out.get("mlm_loss") is not None:
        res["loss_mlm"] = out["mlm_loss"]
        total = total + res["loss_mlm"]
    elif out["mlm_logits"] is not None:
        res["loss_mlm"] = F.cross_entropy(out["mlm_logits"].reshape(-1, 30522).float(), batch["mlm_labels"].to(dev).reshape(-1), ignore_index=-1)
        total = total + res["loss_mlm"]
    if out["itm_logits"] is not None:
        res["loss_itm"] = F.cross_entropy(out["itm_logits"].reshape(-1, 2).float(), batch["itm_labels"].to(dev).reshape(-1))
        total = total + res["loss_itm"]
    if out["sup_cls_logits"] is not None:
        res["loss_sup_cls"] = F.cross_entropy(out["sup_cls_logits"].reshape(-1, 48).float(), batch["sup_cls_labels"].to(dev).reshape(-1))
        res["loss_sub_cls"] = F.cross_entropy(out["sub_cls_logits"].reshape(-1, 122).float(), batch["sub_cls_labels"].to(dev).reshape(-1))
        total = total + res["loss_sup_cls"] + res["loss_sub_cls"]
    if out["t2i_logits"] is not None:
        res["loss_t2i"] = 10 * F.smooth_l1_loss(out["t2i_logits"].float(), batch["image"].to(dev))
        total = total + res["loss_t2i"]
    res["total_loss"] = total
    return res


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", ["tiny256_pretrain", "small96_T20_ragged", "tiny256_ft"])
def test_train_step_parity(golden_dir, name, dtype, fused):
    """one train-mode step (injected dropout / DropPath masks): losses and every parameter gradient vs the oracle,
    and vs the reference's gradient norms in the golden fixture."""
    from tests.golden.make_golden import make_masks
    model, cfg, sd, batch, g, seed = build(name, golden_dir, dtype)
    if fused and not cfg.loss_type["mlm"]:
        pytest.skip("no MLM head")
    dev = torch.device("cuda:0")
    step_idx = 1 if cfg.loss_type["t2i"] else 0
    B, T = batch["image"].shape[0], batch["input_ids"].shape[1]
    masks = make_masks(cfg, B, T, seed + step_idx)
    model.train()
    model.injected_masks = masks
    img = batch["masked_images"] if step_idx == 1 else batch["image"]
    out = model(img.to(dev), batch["input_ids"].to(dev), mlm_labels=batch["mlm_labels"].to(dev) if fused else None)
    ls = _losses_like_engine(out, batch, dev)
    ls["total_loss"].backward()
    torch.cuda.synchronize()
    # oracle
    sdg = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and "running_" not in k) else v)
           for k, v in sd.items() if k != O.TIED[0]}
    if cfg.loss_type["mlm"]:
        sdg[O.TIED[0]] = sdg[O.TIED[1]]
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    lo, _ = O.step_loss(sdg, cfg, batch, step_idx, train=True, masks=masks, bn_out={})
    lo["total_loss"].backward()
    tol = TOL[dtype]
    for k, v in lo.items():
        assert abs(float(ls[k]) - float(v)) <= 2 * tol * max(1.0, abs(float(v))), (k, float(ls[k]), float(v))
    gtol = 5e-3 if dtype == torch.float32 else 8e-2
    worst = {}
    for k, p in model.named_parameters():
        ref_g = sdg[k].grad
        if ref_g is None:
            continue
        assert p.grad is not None, k
        e = ((p.grad.detach().double().cpu() - ref_g.double()).norm() / ref_g.double().norm().clamp_min(1e-12)).item()
        if ref_g.double().norm().item() < 1e-7:
            continue
        worst[k] = e
        gk = f"train{step_idx}/grad/{k}/norm"
        if gk in g.files:
            refn = float(g[gk])
            assert abs(p.grad.double().norm().item() - refn) <= 2 * gtol * max(refn, 1e-6), (k, p.grad.double().norm().item(), refn)
    bad = {k: e for k, e in worst.items() if not e < gtol}
    assert not bad, (name, dtype, sorted(bad.items(), key=lambda kv: -kv[1])[:12])


def test_missing_gpu_path_is_loud():
    from mvlt_amd import pvlt
    m = pvlt.pvlt_tiny(pretrained=False, token_hidden_size=768, num_text_tokens=16,
                       loss_type=dict(mlm=1, itm=1, t2i=0, cls=0), pretrained_pth=None)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 64, 64), torch.zeros(1, 16, dtype=torch.long))
