"""Data-parallel gradient exchange for the MI355X MVLT model: one process per GPU, RCCL over xGMI through
torch.distributed (backend "nccl" IS RCCL on ROCm).

The path shards naturally over the batch: the only exchange step is the gradient all-reduce
(reference: torch DistributedDataParallel's reducer, main_vl.py:298-302).  Here the flat fp32 gradient buffer is
cut into contiguous ranges in gradient-READY order -- heads, stage 4, 3, 2, 1, embeddings -- and each range is
all-reduced (SUM, then the 1/world average is folded in) as soon as the backward schedule has finished writing it.
ProcessGroupNCCL runs collectives on its own HIP stream behind an event on the compute stream, so the transfers
overlap the remaining backward kernels; the xGMI links are point-to-point (7 x ~153 GB/s per GPU) and a ring collective
pays its latency per link hop, so announced ranges are held back until at least MIN_BYTES are pending and neighbouring ranges
travel as one (pvlt_tiny: 8 announcements of 1.5 .. 27 MB become heads 11.7 MB | stage 4 27 MB | stages 3 + 2 18 MB, and at the end
of the pass stage 1 1.5 MB | BERT embeddings 95 MB) instead of DDP's 25 MB + 1 MB first bucket pattern.

Stock torch DistributedDataParallel also works on the model (its reducer hooks fire from the trunk's autograd node);
this wrapper is the overlapped, copy-free path.
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn

# test switch: issue every collective even at world size 1 (a one-GPU box then exercises the RCCL calls of the N > 1 path)
_FORCE = bool(os.environ.get("MVLT_DP_FORCE_COLLECTIVES"))
# A/B + test switch: wait for every collective at the end of the backward and step all parameters in one AdamW launch (round 4's behaviour)
_ONE_LAUNCH = bool(os.environ.get("MVLT_ADAMW_ONE_LAUNCH"))


class DataParallel(nn.Module):
    """`model = DataParallel(model)`; exposes `.module` like DDP (reference main_vl.py:302 reads model.module)."""

    def __init__(self, module, process_group=None, broadcast_buffers=True, grad_payload=None):
        """grad_payload: torch.float32 (default) or torch.bfloat16 -- the dtype the gradient ranges travel in.  bf16 halves the
        bytes on the xGMI ring (80 MB instead of 160 MB per step for pvlt_tiny) at the price of one rounding of
        every rank's partial sum to 8 mantissa bits before the reduction (RCCL then sums in bf16): off unless asked for
        (`MVLT_DP_BF16=1` in the environment, or this argument)."""
        super().__init__()
        self.module = module
        self.pg = process_group
        self.broadcast_buffers = broadcast_buffers
        if grad_payload is None:
            grad_payload = torch.bfloat16 if os.environ.get("MVLT_DP_BF16") else torch.float32
        assert grad_payload in (torch.float32, torch.bfloat16)
        self.grad_payload = grad_payload
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (_FORCE and dist.is_initialized())
        self._works = []
        self._pending = []                      # announced ranges not yet on the wire (see _range_ready)
        self._finishing = False
        self._synced_init = False
        module.store.on_backward_done = self._finish
        module.store.on_range_ready = self._range_ready
        module.store.on_pass_aborted = self._aborted

    # parameters start identical on every rank (DDP broadcasts rank 0's at construction)
    def _sync_init(self):
        if self.active:
            S = self.module.store
            dist.broadcast(S.P, 0, group=self.pg)
            S.force_dirty = True
        self._synced_init = True

    def forward(self, *a, **k):
        if self.active:
            S = self.module.store
            dev = a[0].device
            S.ensure(dev)
            if not self._synced_init:
                self._sync_init()
            if self.broadcast_buffers and self.module.training:
                self._sync_buffers()
        return self.module(*a, **k)

    def _buffer_slab(self):
        """The module's floating-point buffers (BatchNorm running statistics of the MIM decoder: 22 small tensors) as views of ONE flat
        tensor, so that following rank 0 is one public `dist.broadcast` (round 3 went through the private `dist._broadcast_coalesced`).
        Re-pointed lazily: `module.to()` / `.cuda()` replace buffer storage, after which the views are rebuilt; `load_state_dict`
        copies in place and keeps them."""
        items = [(mod, name, b) for mod in self.module.modules() for name, b in mod._buffers.items() if b is not None and b.is_floating_point()]
        if not items:
            return None
        slab = getattr(self, "_slab", None)
        ok = slab is not None
        off = 0
        if ok:
            for _, _, b in items:
                if b.dtype != slab.dtype or b.device != slab.device or b.data_ptr() != slab.data_ptr() + off * slab.element_size():
                    ok = False
                    break
                off += b.numel()
            ok = ok and off == slab.numel()
        if not ok:
            dt, dev = items[0][2].dtype, items[0][2].device
            assert all(b.dtype == dt and b.device == dev for _, _, b in items), "float buffers of one dtype on one device"
            slab = torch.empty(sum(b.numel() for _, _, b in items), dtype=dt, device=dev)
            off = 0
            for mod, name, b in items:
                v = slab[off:off + b.numel()].view(b.shape)
                v.copy_(b)
                mod._buffers[name] = v
                off += b.numel()
            self._slab = slab
        return slab

    def _sync_buffers(self):
        """BatchNorm running stats of the MIM decoder follow rank 0 (DDP's broadcast_buffers=True): one broadcast of the flat slab."""
        if not self.active:
            return
        slab = self._buffer_slab()
        if slab is not None:
            dist.broadcast(slab, 0, group=self.pg)

    def _aborted(self, store):
        """a backward pass raised after announcing ranges: wait for what is in flight and forget it (the next pass starts clean)"""
        for w in [x[0] for x in self._works] + [x[0] for x in store.grad_works]:
            try:
                w.wait()
            except RuntimeError:
                pass
        self._works = []
        store.grad_works = []
        self._pending = []

    MIN_BYTES = 16 << 20                        # pending gradient bytes that start a collective

    @staticmethod
    def _merged(ranges):
        """sorted, with neighbours joined"""
        out = []
        for lo, hi in sorted(ranges):
            if out and out[-1][1] == lo:
                out[-1][1] = hi
            else:
                out.append([lo, hi])
        return [(lo, hi) for lo, hi in out]

    def _flush(self, store):
        for lo, hi in self._merged(self._pending):
            self._reduce(store, lo, hi)
        self._pending = []

    def _reduce(self, store, lo, hi):
        """one collective: (work, lo, hi, G view, payload tensor or None, early) -- early = issued while the backward was still running"""
        g = store.G[lo:hi]
        early = not self._finishing
        if self.grad_payload is torch.float32:
            self._works.append((dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), lo, hi, g, None, early))
        else:
            t = g.to(self.grad_payload)                  # half the bytes on the links; written back (widened) once the collective is done
            self._works.append((dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), lo, hi, g, t, early))

    def _range_ready(self, store, lo, hi):
        """called by the backward schedule when G[lo:hi] is final on the compute stream: queued, and sent once MIN_BYTES are pending"""
        if self.active and hi > lo:
            self._pending.append((lo, hi))
            if sum(h - l for l, h in self._pending) * 4 >= self.MIN_BYTES:
                self._flush(store)

    def _finish(self, store):
        if not self.active:
            return
        store.sync_grads()                                 # MIM-decoder grads that autograd produced itself
        done = getattr(store, "_ranges_done", [])
        # whatever the schedule did not announce (decoder head, padding gaps): reduce the remaining ranges
        cur = 0
        for lo, hi in sorted(done) + [(store.total, store.total)]:
            if lo > cur:
                self._pending.append((cur, lo))
            cur = max(cur, hi)
        self._finishing = True
        try:
            self._flush(store)
        finally:
            self._finishing = False
        # The fused optimizer is the next reader of G (engine.BF16Scaler sets scale_in_optimizer around backward + step): the collectives are handed
        # over un-waited -- FusedAdamW.step steps the ranges whose collectives went out during the backward (pvlt_tiny: 57 of 153 MB) while the tail
        # (the tied word-embedding table, final only with the last kernel of the pass) is still on the wire, then the tail.  Anyone else gets final gradients.
        store.grad_works, self._works = self._works, []
        if not (store.scale_in_optimizer and not _ONE_LAUNCH):
            store.wait_grads()
        store._ranges_done = []
        store.scale_grads(1.0 / self.world)          # DDP's mean; folded into the fused AdamW kernel when that is the optimizer


def allreduce_meter(count, total, device):
    """[count, total] float64 all-reduce used by the metric logger (reference libs/utils.py:38-47)."""
    if not (dist.is_available() and dist.is_initialized()):
        return count, total
    t = torch.tensor([count, total], dtype=torch.float64, device=device)
    dist.barrier()
    dist.all_reduce(t)
    t = t.tolist()
    return int(t[0]), t[1]
