"""Train / eval loops: the MI355X counterpart of reference engine_grid_masking.py.

`train_one_epoch_vl` keeps the reference signature and return value (engine_grid_masking.py:27-32,150) so
main_vl.py:431-437 calls it unchanged.  What differs, on purpose:
  * the MLM loss goes through the model's fused masked-row path (`model(images, ids, mlm_labels=...)`): only rows
    CrossEntropyLoss(ignore_index=-1) keeps are projected onto the 30522-word vocabulary -- same loss value, same
    gradients, no (B, T, 30522) logits tensor.  Models without that keyword get the reference's full-logits path.
  * a forward runs on EVERY iteration (the shipped engine reuses stale outputs on odd iterations when t2i is off
    and crashes in backward -- SURVEY.md App. D #1)
  * the six logged scalars of an iteration travel to the host in ONE asynchronous copy into pinned memory that is read
    only after the backward and the optimizer step of the same iteration have been queued: the host waits for the
    forward alone while the GPU already holds the rest of the step (the reference's six `.item()` calls and its
    torch.cuda.synchronize() each drain the queue, :104-129)
  * the number of selected MLM positions is counted on the host when the loader hands over CPU labels (no device
    round trip); device-resident labels go through the model's asynchronous count (schedule._HostCount)
The compute dtype is a property of the model (bf16 by default, fp32 when `fp32=True`), not an autocast region.
"""
import math

import torch
import torch.nn.functional as F

from ._lib import MVLTError
from .metrics import MetricLogger, SmoothedValue
from .params import pool_zeros

MLM_LOSS_WEIGHT, ITM_LOSS_WEIGHT, T2I_LOSS_WEIGHT = 1, 1, 10      # reference engine_grid_masking.py:23
USE_ORI_INPUT_IDS = False


def _core(model):
    return model.module if hasattr(model, "module") else model


class _SmoothL1Fn(torch.autograd.Function):
    """F.smooth_l1_loss(pred, target) (beta 1, mean) on fp32 device tensors as two HIP passes (sum; gradient): ATen writes and
    re-reads the per-element loss tensor (201 MB at batch 256) before reducing it."""

    @staticmethod
    def forward(ctx, pred, target):
        from . import ops
        pred, target = pred.contiguous(), target.contiguous()
        acc = pool_zeros((1,), torch.float32, pred.device)            # the step's zeroed scratch (valid until the next forward)
        ops.smooth_l1_fwd(pred, target, acc)
        ctx.save_for_backward(pred, target)
        return (acc / pred.numel()).reshape(())

    @staticmethod
    def backward(ctx, gout):
        from . import ops
        pred, target = ctx.saved_tensors
        grad = torch.empty_like(pred)
        ops.smooth_l1_bwd(pred, target, gout.reshape(1).float().contiguous(), grad)
        return grad, None


class _CrossEntropyFn(torch.autograd.Function):
    """CrossEntropyLoss(mean) over the rows of a small fp32 logits matrix (ITM: B x 2, CLS: B x 48 / B x 122; reference
    engine_grid_masking.py:90,94,95) on the HIP row kernels the MLM head uses: one launch forward, one backward, instead of ATen's
    log_softmax / nll_loss pairs."""

    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        from . import ops
        rows, V = logits.shape
        lse = torch.empty(rows, device=logits.device, dtype=torch.float32)
        acc = pool_zeros((2,), torch.float32, logits.device)          # [loss sum, row count]
        ops.cross_entropy_fwd(logits, labels, lse, acc[0:1], acc[1:2], rows, V, V, ignore_index=ignore_index)
        ctx.save_for_backward(logits, labels, lse, acc)
        ctx.ignore_index = ignore_index
        return acc[0] / acc[1]

    @staticmethod
    def backward(ctx, gout):
        from . import ops
        logits, labels, lse, acc = ctx.saved_tensors
        rows, V = logits.shape
        dl = torch.empty_like(logits)
        ops.cross_entropy_bwd(logits, labels, lse, gout.reshape(1).float().contiguous(), acc[1:2], dl, rows, V, V, V, ignore_index=ctx.ignore_index)
        return dl, None, None


def cross_entropy(logits, labels, ignore_index=-100):
    """F.cross_entropy(logits, labels, ignore_index=...) for 2-D logits.  Device tensors ALWAYS take the HIP row kernels (cast to fp32 / int64 first if the
    caller hands something else; anything that is not a 2-D matrix raises): there is no ATen fallback on the GPU.  CPU tensors -- which the model cannot
    produce; they reach this function only when tests/test_host_cpu.py checks the loss composition against the oracle -- take F.cross_entropy."""
    if not logits.is_cuda:
        return F.cross_entropy(logits, labels, ignore_index=ignore_index)
    if logits.dim() != 2 or labels.dim() != 1 or labels.shape[0] != logits.shape[0]:
        raise MVLTError(f"cross_entropy: need (rows, classes) logits and (rows,) labels, got {tuple(logits.shape)} / {tuple(labels.shape)}")
    return _CrossEntropyFn.apply(logits.float().contiguous(), labels.long().contiguous(), ignore_index)


def smooth_l1(pred, target):
    """the T2I loss of reference engine_grid_masking.py:99.  Device tensors always take the two HIP passes (16-byte accesses: the element count must be a
    multiple of 4, which 3 x S x S images with even S are; otherwise it raises); CPU tensors (host-logic test only) take F.smooth_l1_loss."""
    if not pred.is_cuda:
        return F.smooth_l1_loss(pred, target)
    if pred.shape != target.shape or pred.numel() % 4 != 0:
        raise MVLTError(f"smooth_l1: shapes {tuple(pred.shape)} / {tuple(target.shape)} must match and hold a multiple of 4 elements")
    return _SmoothL1Fn.apply(pred.float(), target.float())


class _ComposeFn(torch.autograd.Function):
    """total and [total, w_i * loss_i ...] of up to five fp32 device scalars in one HIP launch (None = head off); backward is one multiply"""
    _w = {}

    @staticmethod
    def forward(ctx, weights, *losses):
        from . import ops
        live = next(l for l in losses if l is not None)
        out = torch.empty(len(LOSS_KEYS), device=live.device, dtype=torch.float32)
        total = torch.empty((), device=live.device, dtype=torch.float32)
        ops.loss_compose([None if l is None else l.detach() for l in losses], weights, out, total)
        ctx.weights, ctx.live = weights, [l is not None for l in losses]
        ctx.set_materialize_grads(False)                 # the parts are only logged in the engine: their gradient arrives as None
        return total, out

    @staticmethod
    def backward(ctx, gtotal, gout):
        ref = gtotal if gtotal is not None else gout
        key = (ref.device, ctx.weights)
        w = _ComposeFn._w.get(key)
        if w is None:
            w = _ComposeFn._w[key] = torch.tensor(ctx.weights, device=ref.device, dtype=torch.float32)
        if gout is None:
            g = gtotal * w                               # d total / d loss_i = w_i
        else:                                            # someone differentiated a part (or the packed total) as well
            g = (gout[1:] + gout[0] + (gtotal if gtotal is not None else 0.0)) * w
        return (None, *[g[i] if live else None for i, live in enumerate(ctx.live)])


def _compose_hip(losses):
    """losses: dict LOSS_KEYS[1:] -> fp32 device scalar or None; -> (total, parts) with `total._packed` = the six values in
    LOSS_KEYS order (what the engine's read-back copies to the host)"""
    weights = (float(MLM_LOSS_WEIGHT), float(ITM_LOSS_WEIGHT), 1.0, 1.0, float(T2I_LOSS_WEIGHT))
    total, packed = _ComposeFn.apply(weights, *[losses[k] for k in LOSS_KEYS[1:]])
    total._packed = packed
    return total, {k: packed[1 + i] for i, k in enumerate(LOSS_KEYS[1:])}


def compute_losses(outputs, images, mlm_labels, itm_labels, sup_cls_labels, sub_cls_labels):
    """Loss composition of reference engine_grid_masking.py:81-102.  Returns (total, dict of the five parts)."""
    dev = images.device
    if images.is_cuda and outputs["mlm_logits"] is None:
        # device path: every loss is one HIP reduction, the composition one more launch
        ls = dict.fromkeys(LOSS_KEYS[1:])
        if outputs.get("mlm_loss") is not None:
            ls["loss_mlm"] = outputs["mlm_loss"].float()
        if outputs["itm_logits"] is not None:
            ls["loss_itm"] = cross_entropy(outputs["itm_logits"].view(-1, 2).float(), itm_labels.view(-1))
        if outputs["sup_cls_logits"] is not None:
            ls["loss_sup_cls"] = cross_entropy(outputs["sup_cls_logits"].view(-1, 48).float(), sup_cls_labels.view(-1))
            ls["loss_sub_cls"] = cross_entropy(outputs["sub_cls_logits"].view(-1, 122).float(), sub_cls_labels.view(-1))
        if outputs.get("t2i_loss") is not None:
            ls["loss_t2i"] = outputs["t2i_loss"].float()
        elif outputs["t2i_logits"] is not None:
            ls["loss_t2i"] = smooth_l1(outputs["t2i_logits"].float(), images)
        if any(v is not None for v in ls.values()):
            return _compose_hip(ls)
    zero = torch.zeros((), device=dev)
    parts = dict(loss_mlm=zero, loss_itm=zero, loss_sup_cls=zero, loss_sub_cls=zero, loss_t2i=zero)
    total = 0
    if outputs.get("mlm_loss") is not None:
        parts["loss_mlm"] = MLM_LOSS_WEIGHT * outputs["mlm_loss"]
        total = total + parts["loss_mlm"]
    elif outputs["mlm_logits"] is not None:
        parts["loss_mlm"] = MLM_LOSS_WEIGHT * cross_entropy(outputs["mlm_logits"].reshape(-1, 30522).float(), mlm_labels.view(-1), ignore_index=-1)
        total = total + parts["loss_mlm"]
    if outputs["itm_logits"] is not None:
        parts["loss_itm"] = ITM_LOSS_WEIGHT * cross_entropy(outputs["itm_logits"].view(-1, 2).float(), itm_labels.view(-1))
        total = total + parts["loss_itm"]
    if outputs["sup_cls_logits"] is not None:
        parts["loss_sup_cls"] = cross_entropy(outputs["sup_cls_logits"].view(-1, 48).float(), sup_cls_labels.view(-1))
        parts["loss_sub_cls"] = cross_entropy(outputs["sub_cls_logits"].view(-1, 122).float(), sub_cls_labels.view(-1))
        total = total + parts["loss_sup_cls"] + parts["loss_sub_cls"]
    if outputs["t2i_logits"] is not None:
        parts["loss_t2i"] = T2I_LOSS_WEIGHT * smooth_l1(outputs["t2i_logits"].float(), images)
        total = total + parts["loss_t2i"]
    return total, parts


def train_step(model, batch, idx, t2i_on, fused=True):
    """forward + losses of engine iteration `idx` on device tensors; returns (total_loss, parts)."""
    images = batch["image"]
    use_masked = (idx % 2 == 1) and t2i_on          # engine_grid_masking.py:72-78
    inp = batch["masked_images"] if use_masked else images
    core = _core(model)
    if fused and hasattr(core, "store") and core.loss_type.get("mlm"):
        outputs = model(inp, batch["input_ids"], mlm_labels=batch["mlm_labels"], mlm_positions=batch.get("mlm_positions"),
                        mlm_count=batch.get("mlm_count"), t2i_target=images if core.loss_type.get("t2i") else None)
    else:
        outputs = model(inp, batch["input_ids"])
    return compute_losses(outputs, images, batch["mlm_labels"], batch["itm_labels"], batch["sup_cls_labels"], batch["sub_cls_labels"])


LOSS_KEYS = ("total_loss", "loss_mlm", "loss_itm", "loss_sup_cls", "loss_sub_cls", "loss_t2i")


class _LossReadback:
    """the six loss scalars of one iteration: device -> pinned host memory, asynchronously, with an event behind the copy"""

    def __init__(self):
        self.pin = torch.empty(len(LOSS_KEYS), dtype=torch.float32).pin_memory()
        self.ev = torch.cuda.Event()

    def post(self, total, parts):
        vals = getattr(total, "_packed", None)             # compute_losses' device path: the six values already sit in one tensor
        if vals is None:
            vals = torch.stack([total.detach().float()] + [parts[k].detach().float() for k in LOSS_KEYS[1:]])
        vals = vals.detach()
        self.pin.copy_(vals, non_blocking=True)
        self.ev.record()

    def get(self):
        self.ev.synchronize()
        return self.pin.tolist()


def to_device_batch(samples, device):
    """The reference's per-tensor `.to(device, non_blocking=True)` (engine_grid_masking.py:42-56) plus the host-side count of
    the MLM selection when the labels arrive on the CPU."""
    keys = ("image", "mlm_labels", "i2t_labels", "masked_images", "itm_labels", "sup_cls_labels", "sub_cls_labels")
    batch = {k: samples[k].to(device, non_blocking=True) for k in keys if k in samples}
    batch["input_ids"] = samples["ori_input_ids" if USE_ORI_INPUT_IDS else "input_ids"].to(device, non_blocking=True)
    if "mlm_positions" in samples:
        batch["mlm_positions"] = samples["mlm_positions"].to(device, non_blocking=True)
    if "mlm_count" in samples:
        batch["mlm_count"] = int(samples["mlm_count"])
    elif "mlm_positions" not in samples and not samples["mlm_labels"].is_cuda:
        batch["mlm_count"] = int((samples["mlm_labels"] != -1).sum())
    return batch


def train_one_epoch_vl(model, criterion, data_loader, optimizer, device, epoch, loss_scaler, max_norm=0,
                       model_ema=None, mixup_fn=None, set_training_mode=True, fp32=False, args=None):
    model.train(set_training_mode)
    core = _core(model)
    if fp32 and getattr(core, "compute_dtype", None) not in (None, torch.float32):
        core.set_compute_dtype(torch.float32)
    logger = MetricLogger(delimiter="  ")
    logger.add_meter("lr", SmoothedValue(window_size=1, fmt="{value:.6f}"))
    header = f"Epoch: [{epoch}]"
    loss_type = getattr(args, "loss_type", None) or core.loss_type
    t2i_on = loss_type.get("t2i", 0) == 1
    readback = _LossReadback()
    for idx, samples in enumerate(logger.log_every(data_loader, 10, header)):
        batch = to_device_batch(samples, device)
        total, parts = train_step(model, batch, idx, t2i_on)
        readback.post(total, parts)
        optimizer.zero_grad()
        is_second_order = hasattr(optimizer, "is_second_order") and optimizer.is_second_order
        loss_scaler(total, optimizer, clip_grad=max_norm, parameters=model.parameters(), create_graph=is_second_order)
        if model_ema is not None:
            model_ema.update(model)
        vals = readback.get()                  # waits for this iteration's forward only; backward + step are already queued
        if not math.isfinite(vals[0]):
            print(f" [ Warning!!! ] Total Loss is {vals[0]} (loss_mlm={vals[1]} | loss_itm={vals[2]} | loss_sup_cls={vals[3]} | "
                  f"loss_sub_cls={vals[4]} | loss_t2i={vals[5]}), non-finite value")
        logger.update(**dict(zip(LOSS_KEYS, vals)))
        logger.update(lr=optimizer.param_groups[0]["lr"])
    logger.synchronize_between_processes(device)
    print("Averaged stats:", logger)
    return {k: m.global_avg for k, m in logger.meters.items()}


train_one_epoch = train_one_epoch_vl      # BASELINE.json's north_star calls it by this name


class BF16Scaler:
    """loss_scaler for bf16 training: the callable / state_dict interface of timm.utils.NativeScaler
    (reference main_vl.py:309,346,426,453; engine_grid_masking.py:126) without fp16 loss scaling, which bf16 does
    not need.  `_scaler` exists because main_vl.py:426 re-assigns it every epoch."""
    state_dict_key = "amp_scaler"

    def __init__(self):
        self._scaler = None

    def __call__(self, loss, optimizer, clip_grad=None, parameters=None, create_graph=False):
        # The data-parallel mean's 1/world may ride in the fused AdamW kernel instead of being a pass over G -- but only HERE, where
        # the next reader of the gradients is known to be that kernel (or the clipping below, which settles the factor first).  Any
        # other loop (timm's NativeScaler, a user's clip_grad_norm_ / logging / stock optimizer) gets final gradients.
        from .optim import FusedAdamW
        store = optimizer.model.store if isinstance(optimizer, FusedAdamW) else None
        if store is not None:
            store.scale_in_optimizer = True
        try:
            loss.backward(create_graph=create_graph)
            # main_vl.py passes --clip-grad (default None = no clipping).  timm's NativeScaler tests `is not None`, so the engine's
            # own default max_norm=0 would scale every gradient to zero there; here 0 means "no clipping" as well.
            if clip_grad:
                assert parameters is not None
                parameters = list(parameters)
                if store is not None:
                    store.apply_pending_scale()                                # clipping reads the gradients: they must be final
                torch.nn.utils.clip_grad_norm_(parameters, clip_grad)
            optimizer.step()
        finally:
            if store is not None:
                store.scale_in_optimizer = False
                store.apply_pending_scale()                                    # nothing owed outside this call (a step that raised)

    def state_dict(self):
        return {}

    def load_state_dict(self, sd):
        pass
