"""Device-side batch preparation and prefetch.

In the reference every sample is masked on the host inside DataLoader workers -- `generate_grid_mask` +
`image.clone().masked_fill_(mask, 1e-6)` (mcloader/fashion_gen.py:167,176,225-254) and `random_masking_features`
(:383-409) -- and the engine then moves seven tensors per batch to the GPU (engine_grid_masking.py:42-56).  Here the loader
only has to deliver the clean image, the original token ids and the three small label tensors:

  DeviceBatchPrep   grid mask (exact-count or the reference's sliding-window generator), the 1e-6 fill, the 80/10/10 token
                    masking and the masked-index selection as HIP kernels (csrc/batchprep.hip) on a counter-based generator
                    (Philox4x32-10 keyed by seed, counted by sample id), bit-exact against oracle/batchprep_oracle.py
  DevicePrefetcher  the reference's DataPrefetcher idea (mcloader/data_prefetcher.py:4-28: copy the NEXT batch on a side stream
                    while the current one computes) for the engine's dict batches: pinned staging, H2D and the preparation
                    kernels on a side HIP stream, the MLM selection count brought back to the host on that stream too, so the
                    step that consumes the batch never waits for it (`mlm_count`, schedule.run_forward)

Both yield the batch-dict schema `train_one_epoch_vl` consumes.
"""
import torch

from . import ops

PATCH = 16                       # fashion_gen.py:167 patch_size=16
FILL = 1e-6                      # fashion_gen.py:176


class DeviceBatchPrep:
    def __init__(self, seed, mask_ratio=0.5, mode="exact", vocab=30522):
        """mode "exact": exactly int(mask_ratio * patches) masked patches per image; "reference": the reference generator's
        sliding-window variant (realised ratio varies around mask_ratio)."""
        assert mode in ("exact", "reference")
        self.seed, self.mask_ratio, self.mode, self.vocab = int(seed), float(mask_ratio), mode, vocab
        self.sample = 0                           # running sample id: sample b of a call is self.sample + b

    def __call__(self, image, ori_input_ids, sample0=None, select=True):
        """image (B,3,S,S) fp32 and ori_input_ids (B,T) int64 on the device -> dict(masked_images, patch_flags, input_ids,
        mlm_labels[, mlm_positions_buf, mlm_count_dev]); kernels go to torch's current stream."""
        assert image.is_cuda and image.dtype == torch.float32 and ori_input_ids.dtype == torch.int64
        image, ori = image.contiguous(), ori_input_ids.contiguous()
        B, C, H, W = image.shape
        gh, gw = H // PATCH, W // PATCH
        if sample0 is None:
            sample0 = self.sample
            self.sample += B
        dev = image.device
        flags = torch.empty(B, gh * gw, dtype=torch.uint8, device=dev)
        ops.grid_mask_flags(flags, B, gh, gw, int(self.mask_ratio * gh * gw), 0 if self.mode == "exact" else 1, self.seed, sample0)
        masked = torch.empty_like(image)
        ops.grid_mask_apply(image, flags, masked, PATCH, FILL)
        ids, labels = torch.empty_like(ori), torch.empty_like(ori)
        ops.token_mask(ori, ids, labels, self.seed, sample0, self.vocab)
        out = dict(masked_images=masked, patch_flags=flags.view(B, gh, gw), input_ids=ids, mlm_labels=labels)
        if select:
            idx = torch.empty(labels.numel(), device=dev, dtype=torch.int32)
            cnt = torch.zeros(1, device=dev, dtype=torch.int32)
            ops.masked_select(labels.view(-1), idx, cnt)
            out["mlm_positions_buf"], out["mlm_count_dev"] = idx, cnt
        return out


class DevicePrefetcher:
    """Iterate `loader` one batch ahead on a side stream.  Every tensor of a sample dict is staged through pinned memory and
    copied asynchronously; with `prep` (a DeviceBatchPrep) the masked image, the masked token ids, the MLM labels and the
    masked-index selection are produced on the device from `image` + `ori_input_ids` instead of being shipped.  Yields dicts
    of device tensors plus `mlm_positions` / `mlm_count` (host int) when the selection was made here."""

    def __init__(self, loader, device, prep=None):
        self.loader, self.device, self.prep = loader, torch.device(device), prep
        self.stream = torch.cuda.Stream(self.device)
        self._pins = {}

    def __len__(self):
        return len(self.loader)

    def _stage(self, key, t):
        if t.is_cuda:
            return t
        pin = self._pins.get(key)
        if pin is None or pin[0].shape != t.shape or pin[0].dtype != t.dtype:
            pin = self._pins[key] = [torch.empty(t.shape, dtype=t.dtype).pin_memory() for _ in range(2)]      # double-buffered
        buf = pin[self._flip]
        buf.copy_(t)
        return buf.to(self.device, non_blocking=True)

    def _preload(self, it):
        try:
            samples = next(it)
        except StopIteration:
            return None
        self._flip ^= 1
        with torch.cuda.stream(self.stream):
            batch = {k: (self._stage(k, v) if isinstance(v, torch.Tensor) else v) for k, v in samples.items()}
            cnt_pin = None
            if self.prep is not None:
                made = self.prep(batch["image"], batch["ori_input_ids"])
                batch.update(masked_images=made["masked_images"], input_ids=made["input_ids"], mlm_labels=made["mlm_labels"],
                             patch_flags=made["patch_flags"])
                idx, cnt = made["mlm_positions_buf"], made["mlm_count_dev"]
            elif "mlm_labels" in batch and "mlm_positions" not in batch:
                lab = batch["mlm_labels"].reshape(-1).contiguous()
                idx = torch.empty(lab.numel(), device=self.device, dtype=torch.int32)
                cnt = torch.zeros(1, device=self.device, dtype=torch.int32)
                ops.masked_select(lab, idx, cnt)
            else:
                idx = cnt = None
            if cnt is not None:
                cnt_pin = self._cnt_pins[self._flip]
                cnt_pin.copy_(cnt, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return batch, idx, cnt_pin, ev

    def __iter__(self):
        self._flip = 0
        self._cnt_pins = [torch.empty(1, dtype=torch.int32).pin_memory() for _ in range(2)]
        it = iter(self.loader)
        nxt = self._preload(it)
        while nxt is not None:
            batch, idx, cnt_pin, ev = nxt
            ev.synchronize()                                   # the side stream finished this batch long ago (one step of lead)
            cur = torch.cuda.current_stream(self.device)
            cur.wait_stream(self.stream)
            for v in batch.values():
                if isinstance(v, torch.Tensor) and v.is_cuda:
                    v.record_stream(cur)
            if idx is not None:
                n = int(cnt_pin[0])
                idx.record_stream(cur)
                batch["mlm_positions"], batch["mlm_count"] = idx[:n], n
            nxt = self._preload(it)                            # next batch starts copying while this one computes
            yield batch
