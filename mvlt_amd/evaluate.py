"""Eval callers of the forward: drop-in counterparts of reference engine_grid_masking.py:153-474 and
of the metric helpers in libs/vl_scores.py.  Same batch keys in, same result keys out, so main_vl.py:467-474 reads
`test_stats["mlm_acc" | "itm_acc" | "sup_cls_acc" | "sub_cls_acc"]` unchanged.

They reuse the HIP forward in eval mode (no saved activations).  What differs from the reference, on purpose:
  * no autocast region: the compute dtype is a property of the model
  * `evaluate_retrieval` / `evaluate_recognition` RETURN their numbers as well as printing them (the reference only prints),
    and `evaluate_recognition` does not write ./visulization/... debug files
  * the F1 / accuracy numbers of `calculate_cls_metrics` are computed here without scikit-learn (same definitions; checked
    against sklearn in tests/test_host_cpu.py)
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .engine import ITM_LOSS_WEIGHT, MLM_LOSS_WEIGHT, T2I_LOSS_WEIGHT
from .metrics import MetricLogger


# ------------------------------------------------------------------ libs/vl_scores.py
def compute_mlm_score(logits, target, index=-1):
    """accuracy of argmax(logits) over the positions with target != index, as a python float (reference
    libs/vl_scores.py:5-34; like the reference it is NaN when nothing is selected: 0 / 0 on tensors)."""
    logits, target = logits.detach(), target.detach()
    preds = logits.argmax(dim=-1)
    keep = target != index
    preds, target = preds[keep], target[keep]
    assert preds.shape == target.shape
    return (torch.sum(preds == target) / target.numel()).item()


def compute_score_with_logits(logits, labels):
    """per-sample 0/1 correctness (reference libs/vl_scores.py:37-51): argmax over dim 1 when there is more than one
    column, sigmoid >= 0.5 against a {0,1} label otherwise."""
    if logits.shape[1] > 1:
        return torch.max(logits, 1)[1] == labels
    return ((torch.sigmoid(logits.reshape(-1)) >= 0.5) == (labels.reshape(-1) == 1)).to(labels.dtype)


def compute_psnr(logits, labels):
    """PSNR with PIXEL_MAX = 255 and no clamping, 100 when the mse is exactly 0 (reference libs/vl_scores.py:54-63)."""
    mse = torch.mean((logits.detach().float() - labels.detach().float()) ** 2).item()
    if mse == 0:
        return 100
    return 20 * math.log10(255.0 / math.sqrt(mse))


# ------------------------------------------------------------------ engine_grid_masking.py:153-333
@torch.no_grad()
def evaluate_vl(data_loader, model, device, args):
    """MLM accuracy on the masked caption, ITM / CLS accuracy on the original caption, MIM PSNR on the grid-masked image, and the
    summed loss.  Every meter is updated on every batch -- with 0 when its head is off -- so the returned dict always has
    mlm_acc, itm_acc, sup_cls_acc, sub_cls_acc, t2i_psnr, total_loss (reference :205-318) -- and "n", the mean batch size, an
    artefact of `metric_logger.update(total_loss=..., n=batch_size)` at :318 that is kept for key-for-key equality."""
    logger = MetricLogger(delimiter="  ")
    model.eval()
    for samples in logger.log_every(data_loader, 10, "Test:"):
        images = samples["image"].to(device, non_blocking=True)
        masked_images = samples["masked_images"].to(device, non_blocking=True)
        mlm_labels = samples["mlm_labels"].to(device, non_blocking=True)
        itm_labels = samples["itm_labels"].to(device, non_blocking=True)
        sup_cls_labels = samples["sup_cls_labels"].to(device, non_blocking=True)
        sub_cls_labels = samples["sub_cls_labels"].to(device, non_blocking=True)
        input_ids = samples["ori_input_ids"].to(device, non_blocking=True)
        input_ids_mlm = samples["input_ids"].to(device, non_blocking=True)
        bs = images.shape[0]
        total = 0.0
        m = dict(mlm_acc=0, itm_acc=0, sup_cls_acc=0, sub_cls_acc=0, t2i_psnr=0)
        out_mlm = model(images, input_ids_mlm)                                       # Part-0 (:200-213)
        if out_mlm["mlm_logits"] is not None:
            total += MLM_LOSS_WEIGHT * F.cross_entropy(out_mlm["mlm_logits"].reshape(-1, 30522).float(), mlm_labels.view(-1), ignore_index=-1).item()
            m["mlm_acc"] = compute_mlm_score(out_mlm["mlm_logits"], mlm_labels)
        out_1 = model(images, input_ids)                                             # Part-I (:221-253)
        if out_1["itm_logits"] is not None:
            lg = out_1["itm_logits"].view(-1, 2).float()
            total += ITM_LOSS_WEIGHT * F.cross_entropy(lg, itm_labels.view(-1)).item()
            m["itm_acc"] = compute_score_with_logits(lg, itm_labels.view(-1)).sum().item() / bs
        if out_1["sup_cls_logits"] is not None:
            sup, sub = out_1["sup_cls_logits"].view(-1, 48).float(), out_1["sub_cls_logits"].view(-1, 122).float()
            total += F.cross_entropy(sup, sup_cls_labels.view(-1)).item() + F.cross_entropy(sub, sub_cls_labels.view(-1)).item()
            m["sup_cls_acc"] = compute_score_with_logits(sup, sup_cls_labels.view(-1)).sum().item() / bs
            m["sub_cls_acc"] = compute_score_with_logits(sub, sub_cls_labels.view(-1)).sum().item() / bs
        if args.loss_type["t2i"] == 1:                                               # Part-III (:295-316)
            out_3 = model(masked_images, input_ids)
            if out_3["t2i_logits"] is None:
                raise Exception("t2i_logits is none, please check the settings!")
            total += T2I_LOSS_WEIGHT * F.smooth_l1_loss(out_3["t2i_logits"].float(), images).item()
            m["t2i_psnr"] = compute_psnr(out_3["t2i_logits"], images)
        for k, v in m.items():
            logger.meters[k].update(v, n=bs)
        logger.update(total_loss=total, n=bs)       # as the reference writes it (:318): this also creates a meter named "n"
    logger.synchronize_between_processes(device)
    print("** mlm@acc {mlm_acc.global_avg:.5f} itm@acc {itm_acc.global_avg:.5f} sup_cls@acc {sup_cls_acc.global_avg:.5f} sub_cls@acc "
          "{sub_cls_acc.global_avg:.5f} t2i@psnr {t2i_psnr.global_avg:.5f} loss {total_loss.global_avg:.5f}".format(**logger.meters))
    return {k: meter.global_avg for k, meter in logger.meters.items()}


# ------------------------------------------------------------------ engine_grid_masking.py:336-393
RETRIEVAL_DENOMINATOR = 1000          # the reference divides the hit counts by a hard-coded 1000 queries (:393)


def rank_candidates(model, images, input_ids):
    """ITM ranking of one query: softmax(itm_logits)[:, 1] per candidate pair, indices sorted by descending score, and the
    position of candidate 0 (the true match) in that order (reference :358-380)."""
    logits = model(images, input_ids)["itm_logits"].view(-1, 2).float()
    score = F.softmax(logits, dim=-1)[:, 1]
    _, order = torch.sort(score, dim=-1, descending=True)
    rank0 = int(np.argwhere(order.cpu().numpy() == 0).reshape(-1)[0])
    return score, order, rank0


@torch.no_grad()
def evaluate_retrieval(data_loader, model, device, args, denominator=RETRIEVAL_DENOMINATOR):
    """acc@1/5/10 of the 101-candidate image<->text retrieval protocol.  Each loader item carries `images_101`
    (1, 101, 3, S, S) and `ori_input_ids_101` (1, 101, T) (reference mcloader/fashion_gen.py:499-505, consumed at
    engine_grid_masking.py:349-350).  `denominator=None` divides by the number of queries seen instead of the reference's 1000."""
    model.eval()
    logger = MetricLogger(delimiter="  ")
    hits = {1: 0, 5: 0, 10: 0}
    n = 0
    for samples in logger.log_every(data_loader, 10, "Test:"):
        images = samples["images_101"].to(device, non_blocking=True).squeeze()
        input_ids = samples["ori_input_ids_101"].to(device, non_blocking=True).squeeze()
        _, _, rank0 = rank_candidates(model, images, input_ids)
        for k in hits:
            hits[k] += int(rank0 < k)
        n += 1
    d = denominator or max(1, n)
    flag = "TIR" if getattr(args, "eval_retrieval_tir", False) else ("ITR" if getattr(args, "eval_retrieval_itr", False) else "")
    res = {f"acc@{k}": v / d for k, v in hits.items()}
    print("\n", "#" * 30, "retrieval evaluation", "#" * 30)
    print(">>> retrieval {}: acc@1: {}, acc@5: {}, acc@10: {}".format(flag, res["acc@1"], res["acc@5"], res["acc@10"]))
    return res


# ------------------------------------------------------------------ engine_grid_masking.py:396-486
def calculate_cls_metrics(cls_labels, preds):
    """(accuracy, macro_f1, micro_f1, weighted_f1) with scikit-learn's definitions (reference :477-486 calls
    sklearn.metrics.f1_score / accuracy_score): classes = union of labels and predictions, per-class F1 = 2tp/(2tp+fp+fn),
    macro = plain mean, weighted = mean weighted by label support, micro = global tp / (tp + (fp+fn)/2) = accuracy here."""
    l, p = np.asarray(cls_labels).reshape(-1), np.asarray(preds).reshape(-1)
    classes = np.union1d(l, p)
    f1, support = [], []
    for c in classes:
        tp = np.sum((p == c) & (l == c))
        fp = np.sum((p == c) & (l != c))
        fn = np.sum((p != c) & (l == c))
        f1.append(2.0 * tp / (2 * tp + fp + fn) if (2 * tp + fp + fn) else 0.0)
        support.append(np.sum(l == c))
    f1, support = np.asarray(f1), np.asarray(support, dtype=np.float64)
    accuracy = float(np.mean(l == p)) if l.size else 0.0
    macro = float(f1.mean()) if f1.size else 0.0
    weighted = float((f1 * support).sum() / support.sum()) if support.sum() else 0.0
    return accuracy, macro, accuracy, weighted


@torch.no_grad()
def evaluate_recognition(data_loader, model, device, args):
    """super- / sub-category recognition: the loader emits `images`, `ori_input_ids`, `sup_cls_labels`, `sub_cls_labels`
    (reference :409-412); predictions are argmax(softmax(logits))."""
    model.eval()
    logger = MetricLogger(delimiter="  ")
    sup_l, sup_p, sub_l, sub_p = [], [], [], []
    for samples in logger.log_every(data_loader, 10, "Test:"):
        images = samples["images"].to(device, non_blocking=True)
        input_ids = samples["ori_input_ids"].to(device, non_blocking=True)
        logits = model(images, input_ids)
        sup_l += list(samples["sup_cls_labels"].view(-1).cpu().numpy())
        sub_l += list(samples["sub_cls_labels"].view(-1).cpu().numpy())
        sup_p += list(torch.max(F.softmax(logits["sup_cls_logits"].view(-1, 48).float(), dim=-1), dim=-1)[1].cpu().numpy())
        sub_p += list(torch.max(F.softmax(logits["sub_cls_logits"].view(-1, 122).float(), dim=-1), dim=-1)[1].cpu().numpy())
    sup, sub = calculate_cls_metrics(sup_l, sup_p), calculate_cls_metrics(sub_l, sub_p)
    print("\n", "#" * 30, "recognition evaluation", "#" * 30)
    print("> logging-sup: accuracy ({}) macro_f1 ({}) micro_f1 ({}) weighted_f1 ({})\n> logging-sub: accuracy ({}) macro_f1 ({}) micro_f1 ({}) "
          "weighted_f1 ({})".format(*sup, *sub))
    keys = ("accuracy", "macro_f1", "micro_f1", "weighted_f1")
    res = {f"sup_{k}": v for k, v in zip(keys, sup)}
    res.update({f"sub_{k}": v for k, v in zip(keys, sub)})
    res.update(sup_cls_preds=[int(v) for v in sup_p], sub_cls_preds=[int(v) for v in sub_p])
    return res


def visual_vl(*args, **kwargs):
    raise NotImplementedError("visual_vl (debug visualisation; reference engine_grid_masking.py:502-685 reads keys the "
                              "dataset no longer emits) is out of scope -- SURVEY.md section 2 row 3")
