"""Fused AdamW over the model's flat fp32 parameter / gradient buffers (one HIP kernel per step, plus the bf16
re-cast of the updated weights in the same pass).

Semantics = torch.optim.AdamW as the reference builds it through timm.optim.create_optimizer (main_vl.py:308):
decoupled weight decay, bias-corrected moments, eps 1e-8, and timm's param-group split -- 1-D tensors and `.bias`
get weight_decay 0, everything else args.weight_decay.  It is a torch.optim.Optimizer, so GradScaler.step(),
lr schedulers (`param_groups[i]['lr']`) and state_dict()/load_state_dict() keep working; `param_groups[0]` is the
no-decay group and `[1]` the decay group, like timm's add_weight_decay.
"""
import os

import torch

from . import ops


def _unwrap(model):
    return model.module if hasattr(model, "module") and not hasattr(model, "store") else model


def phased_ranges(S):
    """The element ranges of the flat buffers in the order the optimizer may step them.  Without gradient collectives in flight: the whole buffer.
    With them (mvlt_amd.dist.DataParallel hands them over un-waited when this optimizer is the next reader): first the ranges whose collectives were
    issued DURING the backward pass -- by now (mostly) complete -- then the ones issued at its end, each group after `wait()` on its works (a
    stream-side wait for RCCL: the host keeps enqueuing).  The first launch then runs while the tail is still on the wire; element-wise AdamW makes the
    result bit-identical to one launch over everything.  Anything the collectives do not cover exactly once falls back to wait-all + one range."""
    works, S.grad_works = S.grad_works, []
    if not works:
        yield 0, S.total
        return
    def finish(ws):
        for w, lo, hi, g, t, early in ws:
            w.wait()
            if t is not None:
                g.copy_(t)
    cov = sorted((lo, hi) for _, lo, hi, _, _, _ in works)
    exact = cov[0][0] == 0 and cov[-1][1] == S.total and all(a[1] == b[0] for a, b in zip(cov, cov[1:]))
    groups = [[x for x in works if x[5]], [x for x in works if not x[5]]]
    if not exact or not groups[0] or not groups[1]:
        finish(works)
        yield 0, S.total
        return
    for ws in groups:
        finish(ws)
        merged = []
        for lo, hi in sorted((x[1], x[2]) for x in ws):
            if merged and merged[-1][1] == lo:
                merged[-1][1] = hi
            else:
                merged.append([lo, hi])
        for lo, hi in merged:
            yield lo, hi


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        model = _unwrap(model)
        self.model = model
        no_decay, decay = [], []
        for name, p in model.named_parameters():
            if not p.requires_grad:
                continue
            (no_decay if (p.dim() == 1 or name.endswith(".bias")) else decay).append(p)
        groups = [dict(params=no_decay, weight_decay=0.0), dict(params=decay, weight_decay=weight_decay)]
        super().__init__(groups, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._step = 0
        self._m = self._v = self._wd_mask = None
        self._hp = self._hp_pin = None
        self._hp_ev = [None] * 4
        self._pending = None                  # (m, v) of a checkpoint loaded before the flat store exists

    def _ensure(self):
        S = self.model.store
        if S.P is None:
            raise RuntimeError("FusedAdamW: run a forward first (the flat parameter store is built lazily)")
        if self._m is None or self._m.numel() != S.total or self._m.device != S.P.device:
            self._m = torch.zeros_like(S.P)
            self._v = torch.zeros_like(S.P)
            if self._pending is not None:        # moments of a checkpoint that was loaded before the first forward
                pm, pv = self._pending
                if pm.numel() != S.total:
                    raise RuntimeError(f"FusedAdamW: checkpoint moments have {pm.numel()} elements, the model's flat store {S.total}")
                self._m.copy_(pm)
                self._v.copy_(pv)
                self._pending = None
            self._hp = torch.zeros(8, device=S.P.device)
            # the step's scalars reach the device through a ring of PINNED rows: from a pageable source hipMemcpyAsync stages the copy on
            # the host and the call returns only when the stream has drained -- the host then enqueues the optimizer kernel and the whole
            # next forward behind an idle GPU (1.3 ms per fine-tune step, tools/host_time.py).  An event per row guards its reuse.
            self._hp_pin = torch.zeros(4, 8).pin_memory() if S.P.is_cuda and not os.environ.get("MVLT_HP_PAGEABLE") else None
            self._hp_ev = [None] * 4
            # one byte per parameter: 1 = weight decay applies (timm's split: not for 1-D tensors / biases)
            ids_nd = {id(p) for p in self.param_groups[0]["params"]}
            mask = torch.zeros(S.total, dtype=torch.uint8)
            for name, p in S.params.items():
                off, n, _ = S.offsets[name]
                if id(p) not in ids_nd:
                    mask[off:off + n] = 1
            self._wd_mask = mask.to(S.P.device)
        return S

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        S = self._ensure()
        S.sync_grads()
        self._step += 1
        g0, g1 = self.param_groups
        b1, b2 = g0["betas"]
        assert g0["lr"] == g1["lr"], "FusedAdamW steps both param groups with one learning rate (as timm's scheduler sets them)"
        gscale, S.pending_grad_scale = S.pending_grad_scale, 1.0      # 1/world of the data-parallel mean, applied in the kernel
        row = [g0["lr"], b1, b2, g0["eps"], g1["weight_decay"], 1 - b1 ** self._step, 1 - b2 ** self._step, gscale]
        if self._hp_pin is None:
            self._hp.copy_(torch.tensor(row, dtype=torch.float32))
        else:
            k = self._step % 4
            if self._hp_ev[k] is not None:
                self._hp_ev[k].synchronize()              # the copy that last used this row has run (four steps ago: a no-op wait)
            self._hp_pin[k].copy_(torch.tensor(row, dtype=torch.float32))
            self._hp.copy_(self._hp_pin[k], non_blocking=True)
            ev = self._hp_ev[k] = self._hp_ev[k] or torch.cuda.Event()
            ev.record()
        for lo, hi in phased_ranges(S):
            ops.adamw_step(S.P[lo:hi], S.G[lo:hi], self._m[lo:hi], self._v[lo:hi], None if S.C is None else S.C[lo:hi], hi - lo, self._hp, self._wd_mask[lo:hi])
        # W^T / permuted conv operand copies are refreshed by the next forward; the plain bf16 copy S.C is already current,
        # which holds as long as nothing else writes the parameters before that forward (FlatStore.versions() notices)
        S.force_dirty = True
        S._c_fresh_version = S.versions() if S.C is not None else None
        return loss

    def zero_grad(self, set_to_none=True):
        super().zero_grad(set_to_none=set_to_none)

    def state_dict(self):
        sd = super().state_dict()
        m, v = (self._m, self._v) if self._m is not None else (self._pending or (None, None))
        sd["fused"] = dict(step=self._step, m=None if m is None else m.cpu(), v=None if v is None else v.cpu())
        return sd

    def load_state_dict(self, sd):
        """Works before the first forward too (the reference resumes in that order: main_vl.py:308 builds the optimizer,
        :340 loads its state, the first forward comes later): the moments then wait in `_pending` until `_ensure`."""
        sd = dict(sd)                                  # the caller's checkpoint dict stays as it was
        fused = sd.pop("fused", None)
        super().load_state_dict(sd)
        if fused is not None:
            self._step = fused["step"]
            if fused["m"] is not None:
                if self.model.store.P is not None:
                    self._pending = None
                    self._ensure()
                    self._m.copy_(fused["m"])
                    self._v.copy_(fused["v"])
                else:
                    self._m = self._v = None
                    self._pending = (fused["m"].detach().clone(), fused["v"].detach().clone())
