"""Eval callers of the forward: counterparts of reference engine_grid_masking.py:153-474.
They reuse the same HIP forward (eval mode: no saved activations) and restate the reference's cheap metrics
(libs/vl_scores.py) in torch on the device."""
import torch
import torch.nn.functional as F

from .metrics import MetricLogger


def compute_mlm_score(logits, labels):
    """accuracy over positions with label != -1 (reference libs/vl_scores.py:5-34)."""
    pred = logits.argmax(dim=-1)
    keep = labels != -1
    n = keep.sum().clamp_min(1)
    return ((pred == labels) & keep).sum().float() / n


def compute_score_with_logits(logits, labels):
    """top-1 accuracy (reference libs/vl_scores.py:37-51)."""
    return (logits.argmax(dim=-1).view(-1) == labels.view(-1)).float().mean()


def compute_psnr(a, b, max_val=1.0):
    """PSNR of two image batches in [0,1] (reference libs/vl_scores.py:54-63)."""
    mse = F.mse_loss(a.float(), b.float())
    return 10.0 * torch.log10(max_val ** 2 / mse.clamp_min(1e-12))


@torch.no_grad()
def evaluate_vl(data_loader, model, device, args):
    """MLM / ITM / CLS accuracy and MIM PSNR over a loader (reference engine_grid_masking.py:153-333)."""
    model.eval()
    logger = MetricLogger(delimiter="  ")
    lt = args.loss_type
    for samples in logger.log_every(data_loader, 10, "Test:"):
        images = samples["image"].to(device, non_blocking=True)
        ids = samples["input_ids"].to(device, non_blocking=True)
        ori = samples.get("ori_input_ids", samples["input_ids"]).to(device, non_blocking=True)
        labels = samples["mlm_labels"].to(device, non_blocking=True)
        out = model(images, ids)
        if lt.get("mlm") and out["mlm_logits"] is not None:
            logger.update(mlm_acc=compute_mlm_score(out["mlm_logits"], labels).item())
        out2 = model(images, ori)
        if lt.get("itm") and out2["itm_logits"] is not None:
            logger.update(itm_acc=compute_score_with_logits(out2["itm_logits"].view(-1, 2), samples["itm_labels"].to(device)).item())
        if lt.get("cls") and out2["sup_cls_logits"] is not None:
            logger.update(sup_acc=compute_score_with_logits(out2["sup_cls_logits"].view(-1, 48), samples["sup_cls_labels"].to(device)).item())
            logger.update(sub_acc=compute_score_with_logits(out2["sub_cls_logits"].view(-1, 122), samples["sub_cls_labels"].to(device)).item())
        if lt.get("t2i") and "masked_images" in samples:
            out3 = model(samples["masked_images"].to(device, non_blocking=True), ori)
            if out3["t2i_logits"] is not None:
                logger.update(t2i_psnr=compute_psnr(out3["t2i_logits"].clamp(0, 1), images).item())
    logger.synchronize_between_processes(device)
    print("* " + str(logger))
    return {k: m.global_avg for k, m in logger.meters.items()}


@torch.no_grad()
def evaluate_retrieval(data_loader, model, device, args, denominator=None):
    """ITM ranking over 101 candidates per query: hit@1/5/10 of candidate 0 (reference engine_grid_masking.py:336-393;
    the reference divides by a hard-coded 1000, App. D #8 -- here by the number of queries unless `denominator`)."""
    model.eval()
    hits = {1: 0, 5: 0, 10: 0}
    n = 0
    for samples in data_loader:
        if "images_101" in samples:                     # image retrieval: one caption, 101 images
            images = samples["images_101"].squeeze(0).to(device)
            ids = samples["input_ids"].to(device).expand(images.shape[0], -1).contiguous()
        else:                                           # text retrieval: one image, 101 captions
            ids = samples["input_ids_101"].squeeze(0).to(device)
            images = samples["image"].to(device).expand(ids.shape[0], -1, -1, -1).contiguous()
        score = model(images, ids)["itm_logits"].view(-1, 2).float().softmax(-1)[:, 1]
        rank0 = int((score.argsort(descending=True) == 0).nonzero()[0])
        for k in hits:
            hits[k] += int(rank0 < k)
        n += 1
    d = denominator or max(1, n)
    res = {f"acc@{k}": v / d for k, v in hits.items()}
    print(res)
    return res


@torch.no_grad()
def evaluate_recognition(data_loader, model, device, args):
    """super-/sub-category accuracy and macro-F1 (reference engine_grid_masking.py:396-474)."""
    model.eval()
    ps, ls, pb, lb = [], [], [], []
    for samples in data_loader:
        out = model(samples["image"].to(device), samples["input_ids"].to(device))
        ps.append(out["sup_cls_logits"].view(-1, 48).float().softmax(-1).argmax(-1).cpu())
        pb.append(out["sub_cls_logits"].view(-1, 122).float().softmax(-1).argmax(-1).cpu())
        ls.append(samples["sup_cls_labels"].view(-1).cpu())
        lb.append(samples["sub_cls_labels"].view(-1).cpu())
    ps, ls, pb, lb = map(torch.cat, (ps, ls, pb, lb))

    def macro_f1(p, l, n):
        f = []
        for c in range(n):
            tp = ((p == c) & (l == c)).sum().item()
            fp = ((p == c) & (l != c)).sum().item()
            fn = ((p != c) & (l == c)).sum().item()
            if tp + fp + fn:
                f.append(2 * tp / (2 * tp + fp + fn))
        return sum(f) / max(1, len(f))

    res = dict(sup_acc=(ps == ls).float().mean().item(), sup_macro_f1=macro_f1(ps, ls, 48),
               sub_acc=(pb == lb).float().mean().item(), sub_macro_f1=macro_f1(pb, lb, 122))
    print(res)
    return res


def visual_vl(*args, **kwargs):
    raise NotImplementedError("visual_vl (debug visualisation; reference engine_grid_masking.py:502-685 reads keys the "
                              "dataset no longer emits) is out of scope -- SURVEY.md section 2 row 3")
