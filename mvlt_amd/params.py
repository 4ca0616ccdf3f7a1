"""Parameter holders and the flat parameter / gradient store of the MI355X MVLT model.

The module tree below exists only to give every tensor the reference's state_dict name and shape
(reference libs/pvlt.py:200-277, libs/vl_heads.py).  It has no forward of its own: the
model's forward drives HIP kernels directly on the flat buffers.

Memory layout (sized for 288 GB HBM: everything stays resident)
  P   fp32 master parameters, one flat buffer, every tensor at an 8-element (32 B) aligned offset
  G   fp32 gradients, same offsets; kernels accumulate into it with atomics (zeroed once per step)
  C16 bf16 copy of P (same offsets) when the compute dtype is bf16; weights additionally get a transposed
      copy W^T (dgrad operand) and conv weights a [out][kh][kw][cin] re-ordering (patch-GEMM operand)
"""
import math
import os
import weakref
from collections import OrderedDict

import torch
import torch.nn as nn

from . import ops

_EARLY_FOLDS = bool(os.environ.get("MVLT_EARLY_FOLDS"))        # A/B switch: gradient-copy folds after every backward stage also without a data-parallel wrapper
_TN_SCRATCH_MIB = int(os.environ.get("MVLT_TN_SCRATCH_MIB", "256"))      # scratch of the weight-gradient partial tiles (deferred folds keep several launches' tiles in it)
_PREP_FP32_SRC = bool(os.environ.get("MVLT_PREP_FP32_SRC"))     # A/B switch: transposed weight copies read the fp32 masters (rounds 1-3)
ALIGN = 8


class Holder(nn.Module):
    """A named bag of parameters (no forward)."""

    def __init__(self, **shapes):
        super().__init__()
        for name, shape in shapes.items():
            self.register_parameter(name, nn.Parameter(torch.zeros(*shape)))


def affine(n):          # LayerNorm-like holder
    return Holder(weight=(n,), bias=(n,))


def linear(n_out, n_in, bias=True):
    return Holder(weight=(n_out, n_in), bias=(n_out,)) if bias else Holder(weight=(n_out, n_in))


def trunc_normal_(t, std=0.02):
    # timm==0.3.2 trunc_normal_: N(0, std) truncated to the ABSOLUTE interval [-2, 2]
    return nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)


def conv_default_init_(w, b):
    # nn.Conv2d.reset_parameters (the reference leaves convs at PyTorch defaults, libs/pvlt.py:282-289)
    nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    if b is not None:
        fan_in = w[0].numel()
        bound = 1 / math.sqrt(fan_in)
        nn.init.uniform_(b, -bound, bound)


class ZeroPool:
    """Per-step scratch that must start at zero (atomic accumulation targets: weight-gradient tiles, BN statistics, dK/dV,
    split-K partial sums ...).  One `zero_()` of a pooled buffer at the start of a step replaces ~50 tiny fill launches;
    a request that does not fit falls back to torch.zeros and grows the pool for the next step.  Tensors taken from the
    pool are valid until the next `reset()` (= next training/eval step): only use it for schedule-internal scratch."""
    _pools = {}

    class Token:
        """held by the autograd node of a forward that took pooled scratch: alive = that forward's backward is still pending"""
        __slots__ = ("__weakref__",)

    def __init__(self, device):
        self.device, self.buf, self.used, self.need = device, None, 0, 0
        self._live = weakref.WeakSet()   # tokens of forwards whose backward has not run (and whose graph has not been freed) yet

    @classmethod
    def of(cls, device):
        key = (device.type, device.index)
        if key not in cls._pools:
            cls._pools[key] = cls(device)
        return cls._pools[key]

    @property
    def pending(self):
        return len(self._live)

    def reset(self, grad_on=False):
        """Start a new step; returns a token when `grad_on` (the caller parks it on its autograd node: the pool counts a forward as
        pending for exactly as long as that node lives -- until its backward has run or its graph was dropped).  While an earlier
        forward is pending (gradient accumulation over micro-batches, an eval / EMA forward between forward and backward) its pooled
        tensors must survive: the buffer is then left to them (they hold the storage) and this step draws from a fresh one of the
        size a step needs (never sized from the abandoned buffer: that grew geometrically when a forward never got its backward)."""
        want = max(self.need, self.used)
        if self.pending > 0:
            self.buf = None
        tok = None
        if grad_on:
            tok = ZeroPool.Token()
            self._live.add(tok)
        if self.buf is None or want > self.buf.numel():
            self.buf = torch.empty(int(want * 1.25) + (1 << 20), dtype=torch.uint8, device=self.device)
            want = self.buf.numel()                              # a fresh buffer is zeroed in full
        if want:
            self.buf[:want].zero_()
        self.used, self.need = 0, 0
        return tok

    def take(self, shape, dtype):
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = (n * torch.empty((), dtype=dtype).element_size() + 255) // 256 * 256
        self.need += nbytes
        if self.buf is None or self.used + nbytes > self.buf.numel():
            return torch.zeros(shape, dtype=dtype, device=self.device)
        t = self.buf[self.used:self.used + nbytes].view(dtype)[:n].view(shape)
        self.used += nbytes
        return t


def pool_zeros(shape, dtype, device):
    return ZeroPool.of(torch.device(device) if not isinstance(device, torch.device) else device).take(tuple(shape) if not isinstance(shape, int) else (shape,), dtype)


class FlatStore:
    """Flat fp32 master/grad buffers behind a module's parameters, plus compute-dtype operand copies."""

    def __init__(self, module, compute_dtype):
        self.module = module
        self.compute_dtype = compute_dtype
        self.device = None
        self.P = self.G = self.C = None
        self._views = {}
        self.offsets = OrderedDict()      # name -> (offset, numel, shape)
        self.params = OrderedDict()       # name -> Parameter (unique)
        self.total = 0
        self._cast_version = -1
        self.refresh_count = 0                   # bumped whenever refresh() rebuilds the derived copies
        self._c_fresh_version = None
        self.extra = {}                   # derived operand copies: name -> tensor
        self._finalize_queued = False
        self.force_dirty = True
        self.fn_params = []
        self.on_backward_done = None      # optional callable(store): data-parallel gradient exchange hook
        self.on_range_ready = None        # optional callable(store, lo, hi): G[lo:hi] is final (overlapped all-reduce)
        self.on_pass_aborted = None       # optional callable(store): a backward pass died before its final callback (see new_pass)
        self._ranges_done = []
        self.all_zeroed_this_pass = False        # set by begin_backward: the whole gradient buffer was zeroed for the running pass
        self.touched_this_pass = set()           # gradient slices a kernel of the running pass has written (first writers may store instead of add)
        self.scale_in_optimizer = False   # set by engine.BF16Scaler around backward + FusedAdamW.step: that kernel applies `pending_grad_scale`
        self.pending_grad_scale = 1.0     # factor still owed to G (1/world after a data-parallel SUM all-reduce)
        # gradient collectives still in flight when the backward pass returned: [(work, lo, hi, g_view, payload or None, early)].  Only set while
        # `scale_in_optimizer` (the fused optimizer is the next reader: it steps the ranges whose collectives are done while the tail is on the
        # wire, FusedAdamW.step); every other reader of G goes through wait_grads() first.
        self.grad_works = []

    # ------------------------------------------------------------------ layout
    def _index(self):
        self.offsets.clear()
        self.params.clear()
        off = 0
        seen = set()
        for name, p in self.module.named_parameters():      # de-duplicates the tied decoder weight
            if id(p) in seen:
                continue
            seen.add(id(p))
            n = p.numel()
            self.offsets[name] = (off, n, tuple(p.shape))
            self.params[name] = p
            off += (n + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        # parameters whose gradients the HIP schedule writes itself: all of them
        self.fn_params = list(self.params.items())

    def is_current(self):
        if self.P is None:
            return False
        for name in (next(iter(self.params)), next(reversed(self.params))):
            off, n, _ = self.offsets[name]
            if self.params[name].data_ptr() != self.P.data_ptr() + 4 * off:
                return False
        return True

    def materialize(self, device):
        """(Re)build the flat buffers on `device` from the current parameter values and re-point every
        Parameter at its slice.  Called lazily: nn.Module.to()/.cuda() replace parameter storage."""
        self._index()
        P = torch.zeros(self.total, device=device, dtype=torch.float32)
        for name, p in self.params.items():
            off, n, shape = self.offsets[name]
            P[off:off + n].copy_(p.detach().reshape(-1).to(device=device, dtype=torch.float32))
        self.P = P
        self.G = torch.zeros_like(P)
        for name, p in self.params.items():
            off, n, shape = self.offsets[name]
            p.data = P[off:off + n].view(shape)
            p.grad = None
        self.C = torch.empty(self.total, device=device, dtype=torch.bfloat16) if self.compute_dtype == torch.bfloat16 else None
        self.device = device
        self.extra = {}
        self._cast_version = -1
        self._c_fresh_version = None
        self._plist = list(self.params.values())
        self.force_dirty = True

    def ensure(self, device):
        if self.device != device or not self.is_current():
            self.materialize(device)

    # ------------------------------------------------------------------ views
    # (the views are cached per flat buffer: a step asks for ~1000 of them, and slicing + view cost ~1 us each on the host -- tools/host_profile.py)
    def _view(self, kind, buf, name):
        cache = self._views.get(kind)
        if cache is None or cache[0] is not buf:
            cache = self._views[kind] = (buf, {})
        v = cache[1].get(name)
        if v is None:
            off, n, shape = self.offsets[name]
            v = cache[1][name] = buf[off:off + n].view(shape)
        return v

    def master(self, name):
        return self._view("P", self.P, name)

    def grad(self, name):
        # NOT cached: the backward hands these views to autograd, whose AccumulateGrad adopts a returned tensor as `.grad` only when nobody else holds it --
        # a cached (shared) view is cloned instead, and `.grad` then no longer aliases G (234 copies per step in sync_grads, and the accumulate / zero
        # logic of begin_backward sees foreign tensors)
        off, n, shape = self.offsets[name]
        return self.G[off:off + n].view(shape)

    def comp(self, name):
        """parameter in the compute dtype, reference layout"""
        return self._view("C", self.C if self.C is not None else self.P, name)

    # ------------------------------------------------------------------ operand copies
    def versions(self):
        """Write counter of the master parameters.  After materialize() every Parameter is its own view of P with its own
        version counter: load_state_dict, torch.optim steps, EMA copies and p.mul_() bump the Parameter's, collectives and
        flat-buffer ops on P bump P's -- so both are read (a few hundred integer attribute reads per forward).  Writers that
        go through the C ABI (FusedAdamW) bump neither and set force_dirty."""
        return self.P._version + sum(p._version for p in self._plist)

    def refresh(self, transposed, conv_perm, conv3=()):
        """Bring compute-dtype copies up to date with the fp32 masters.
        transposed: names of 2-D weights needing W^T; conv_perm: names of kernel==stride conv weights used as patch
        GEMMs; conv3: names of the MIM decoder's 3x3 conv weights (forward taps + flipped/transposed dgrad taps)."""
        ver = self.versions()
        if not self.force_dirty and self._cast_version == ver:
            return
        if self.C is not None and self._c_fresh_version != ver:
            ops.cast_bf16(self.P, self.C, self.total)        # (the fused AdamW step writes the bf16 copy itself: skipped then)
        self._c_fresh_version = None
        key = (tuple(transposed), tuple(conv_perm), tuple(conv3), self.compute_dtype, self.P.data_ptr())
        if getattr(self, "_prep_key", None) != key:
            self._build_prep(transposed, conv_perm, conv3)
            self._prep_key = key
        if self._prep_n:
            ops.weight_prep(self._prep_desc, self._prep_blk, self._prep_n, self._prep_blocks, self.compute_dtype, self._prep_blk_desc)
        self._cast_version = ver
        self.force_dirty = False
        self.refresh_count += 1      # consumers that cache their own derived copies key on this

    def _build_prep(self, transposed, conv_perm, conv3):
        """Descriptor table of every derived weight copy (one mvlt_weight_prep launch per step refreshes them all).
        The destination tensors are allocated once here; their addresses are baked into the table."""
        import ctypes
        from ._lib import PrepDesc
        dt, dev = self.compute_dtype, self.device
        descs, blocks = [], []

        def src_ptr(name):
            return self.master(name).data_ptr()

        def transpose(name):
            w = self.master(name)
            R, Ccols = w.shape
            ld = (R + 7) // 8 * 8                            # rows of W^T padded to 16 B (vocab 30522 -> 30528)
            k = name + "::T"
            if k not in self.extra or self.extra[k].dtype != dt or self.extra[k].device != dev:
                self.extra[k] = torch.zeros(Ccols, ld, device=dev, dtype=dt)
            if self.C is not None and dt == torch.bfloat16 and Ccols % 8 == 0 and not _PREP_FP32_SRC:
                # W^T from the bf16 copy of the parameters (fresh whenever this launch runs: refresh() casts first, or the fused optimizer step wrote
                # it): identical values -- bf16(W)^T -- at half the bytes read, in 16-byte accesses
                descs.append(PrepDesc(self.comp(name).data_ptr(), self.extra[k].data_ptr(), 2, R, Ccols, ld, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0))
            else:
                descs.append(PrepDesc(src_ptr(name), self.extra[k].data_ptr(), 0, R, Ccols, ld, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0))
            blocks.append(((R + 63) // 64) * ((Ccols + 63) // 64))

        def gather(name, suffix, shape, dims, src_off, ss, ds):
            k = name + suffix
            if k not in self.extra or self.extra[k].dtype != dt or tuple(self.extra[k].shape) != tuple(shape) or self.extra[k].device != dev:
                self.extra[k] = torch.zeros(*shape, device=dev, dtype=dt)
            descs.append(PrepDesc(src_ptr(name), self.extra[k].data_ptr(), 1, 0, 0, 0, dims[0], dims[1], dims[2], src_off,
                                  ss[0], ss[1], ss[2], ds[0], ds[1], ds[2]))
            blocks.append((dims[0] * dims[1] * dims[2] + 255) // 256)

        for name in transposed:
            transpose(name)
        for name in list(conv_perm) + list(conv3):
            out, cin, kh, kw = self.master(name).shape
            t = kh * kw
            # [out][kh][kw][cin]: dst (out, t, cin) <- src[out][cin][t]
            gather(name, "::K", (out, t * cin), (out, t, cin), 0, (cin * t, 1, t), (t * cin, cin, 1))
            if name in conv_perm:
                # transpose of ::K: dst (t, cin, out) <- src[out][cin][t]
                gather(name, "::KT", (t * cin, out), (t, cin, out), 0, (1, t, cin * t), (cin * out, out, 1))
            else:
                # dgrad taps: dst (cin, 8 - t, out) <- src[out][cin][t]   ([cin][2-dy][2-dx][out])
                gather(name, "::F", (cin, t * out), (cin, t, out), t - 1, (t, -1, cin * t), (t * out, out, 1))
        if "t2i_head.score.0.weight" in self.offsets and conv3:
            name = "t2i_head.score.0.weight"
            K = self.master(name).numel() // 3
            gather(name, "::W", (3, K), (1, 1, 3 * K), 0, (0, 0, 1), (0, 0, 1))
            gather(name, "::T", (K, 8), (K, 3, 1), 0, (1, K, 0), (8, 1, 0))            # W^T, rows padded to 8 (zeros stay)
        self._prep_n = len(descs)
        if not descs:
            return
        arr = (PrepDesc * len(descs))(*descs)
        raw = torch.frombuffer(bytearray(ctypes.string_at(ctypes.addressof(arr), ctypes.sizeof(arr))), dtype=torch.uint8)
        self._prep_desc = raw.to(dev)
        starts = [0]
        for b in blocks:
            starts.append(starts[-1] + b)
        self._prep_blk = torch.tensor(starts, dtype=torch.int32, device=dev)
        self._prep_blocks = starts[-1]
        # block -> descriptor (the inverse of the prefix sums): the kernel's workgroups look themselves up with two scalar loads
        self._prep_blk_desc = torch.repeat_interleave(torch.arange(len(blocks), dtype=torch.int32), torch.tensor(blocks)).to(torch.int32).to(dev)

    # ------------------------------------------------------------------ gradients
    def begin_backward(self):
        """First HIP-scheduled node of a backward pass: every kernel of the pass ACCUMULATES into G, so G must hold what the
        caller's `.grad` tensors stand for at this moment -- zeros when they are None (optimizer.zero_grad(), the reference
        order forward -> zero_grad -> backward, engine_grid_masking.py:40-127) or foreign tensors (autograd adds the slices we
        return to those), the running sum when they alias G (accumulation over several backward passes without zero_grad;
        zero_grad(set_to_none=False) zeroes G through the alias).  Deciding here and not in the forward is what makes the
        second and later optimizer steps correct: during the forward `.grad` still aliases G from the previous step."""
        # accumulate only if a trainable parameter's .grad aliases its slice of G; frozen parameters (requires_grad=False, .grad stays
        # None for ever) and parameters the optimizer's zero_grad does not cover say nothing about the caller's intent
        self.wait_grads()                 # collectives of a pass nobody stepped (no-op normally)
        if self.G.is_cuda:
            from . import ops
            if getattr(self, "_tn_partials", None) is not None:
                ops.tn_fold_discard(self._tn_partials)         # THIS store's folds (the table is kept per scratch) a pass that raised left pending (nothing, normally): dropped, never run -- their gradients are void, their buffers may be gone
        live = [(n, q) for n, q in self.fn_params if q.requires_grad]
        alias = [q.grad is not None and q.grad.data_ptr() == self.grad(n).data_ptr() for n, q in live]
        self.all_zeroed_this_pass = not any(alias)     # G is all zeros now: a pass's FIRST writer of a slice may store instead of add (schedule._mlm_decoder_bwd)
        self.touched_this_pass = set()
        if not any(alias):
            self.G.zero_()
            self.pending_grad_scale = 1.0
        else:
            # mixed state (a zero_grad over a subset of the parameters, or p.grad = None on some): decided per parameter, like torch --
            # the slices whose .grad aliases keep their running sum, the others start from zero (the any-alias rule of round 3
            # handed the previous step's sums out again for the parameters that had been reset).  Neighbouring slices merge into one fill.
            self.apply_pending_scale()
            runs = []
            for (n, _), a in zip(live, alias):
                if a:
                    continue
                off, cnt, _ = self.offsets[n]
                hi = off + (cnt + ALIGN - 1) // ALIGN * ALIGN
                if runs and runs[-1][1] == off:
                    runs[-1][1] = hi
                else:
                    runs.append([off, hi])
            for lo, hi in runs:
                self.G[lo:hi].zero_()
        self._ranges_done = []
        self._own = set()
        if getattr(self, "_tap_lo", None) is not None:       # a backward that died between a conv weight-gradient GEMM and its fold
            self._tap_arena.zero_()
            self._tap_lo = self._tap_hi = None
        if getattr(self, "_ln_lo", None) is not None:        # same for the LayerNorm accumulator copies of a pass that never folded
            self._ln_arena[:, self._ln_lo:self._ln_hi].zero_()
            self._ln_lo = self._ln_hi = None

    def own(self, t):
        """register a gradient tensor a backward node of this pass created itself: later nodes of the same pass may then accumulate
        into it in place instead of cloning (autograd hands gradients over by reference; tensors of unknown origin are never written)"""
        self._own.add(t.data_ptr())
        return t

    def owns(self, t):
        return t is not None and t.data_ptr() in getattr(self, "_own", ())

    def new_pass(self):
        """Called by every grad-enabled forward.  `_finalize_queued` is cleared by the autograd engine's final callback; a backward
        that raised (OOM, a kernel check, KeyboardInterrupt) drops its queued callbacks, and the flag would stay set for the life of
        the store: every later backward would then skip `begin_backward` (G never zeroed, ranges never reset) and `_finalize` (no
        fold, no data-parallel exchange).  No forward runs inside a backward pass, so a set flag here IS a dead pass."""
        if self._finalize_queued:
            self._finalize_queued = False
            self._ranges_done = []
            if self.on_pass_aborted is not None:
                self.on_pass_aborted(self)

    def queue_finalize(self):
        """Called at the top of every HIP-scheduled backward node: the first call of a backward pass prepares G
        (`begin_backward`) and queues `_finalize` to run once when this pass ends."""
        if self._finalize_queued:
            return
        self._finalize_queued = True
        self.begin_backward()
        torch.autograd.Variable._execution_engine.queue_callback(self._finalize)

    def _finalize(self):
        self._finalize_queued = False
        self.fold_copies()
        if self.on_backward_done is not None:
            self.on_backward_done(self)

    def stage_range(self, i):
        """[lo, hi) of the flat buffers holding patch_embed{i+1}, text_embed{i+1} and block{i+1} (contiguous)."""
        names = [n for n in self.offsets if n.startswith((f"patch_embed{i+1}.", f"text_embed{i+1}.", f"block{i+1}."))]
        lo = self.offsets[names[0]][0]
        o, n, _ = self.offsets[names[-1]]
        return lo, o + (n + ALIGN - 1) // ALIGN * ALIGN

    # ---- interleaved accumulators for the LayerNorm parameter gradients ------------------------------------------------
    # Every workgroup of a LayerNorm backward adds its 2*C partial sums to the same few cache lines; those atomics serialise at
    # the memory side (16-27 us per launch with one accumulator; still 12 / 20 / 29 us at C = 320 / 512 / 768 with 8 interleaved copies, round 4).
    # The launches therefore add into LN_COPIES copies inside an arena -- ONE PER WORKGROUP (the launch has at most 256), so the adds need no
    # atomics at all -- and one `mvlt_fold_copies` launch per backward stage sums the copies into G (and zeroes the ones that were written).
    # Arena slots are handed out in first-use order, so a stage's slots are one contiguous range; a copy has room for every 1-D parameter of the
    # model, whatever its depth (pvlt_tiny: 256 x 56 k floats = 57 MB).
    LN_COPIES = int(os.environ.get("MVLT_LN_COPIES", "256"))

    @property
    def ln_stride(self):
        """floats per accumulator copy: room for every 1-D parameter of the model (all LayerNorm weights / biases are among them)"""
        if getattr(self, "_ln_stride", None) is None:
            total = sum(n for _, n, shape in self.offsets.values() if len(shape) == 1)
            self._ln_stride = (total + 63) // 64 * 64
        return self._ln_stride

    def ln_kwargs(self):
        return dict(copies=self.LN_COPIES, copy_stride=self.ln_stride)

    def grad_copies(self, name):
        """arena view (copy 0) standing in for self.grad(name) in ops.layernorm_bwd(..., **self.ln_kwargs())"""
        if getattr(self, "_ln_arena", None) is None or self._ln_arena.device != self.G.device:
            self._ln_arena = torch.zeros(self.LN_COPIES, self.ln_stride, device=self.G.device, dtype=torch.float32)
            self._ln_index = torch.zeros(self.ln_stride, device=self.G.device, dtype=torch.int32)
            self._ln_slots, self._ln_next, self._ln_lo, self._ln_hi = {}, 0, None, None
        goff, n, _ = self.offsets[name]
        slot = self._ln_slots.get(name)
        if slot is None:
            assert self._ln_next + n <= self.ln_stride, "LayerNorm gradient arena too small"
            slot = self._ln_slots[name] = self._ln_next
            self._ln_index[slot:slot + n] = torch.arange(goff, goff + n, device=self.G.device, dtype=torch.int32)
            self._ln_next += n
        self._ln_lo = slot if self._ln_lo is None else min(self._ln_lo, slot)
        self._ln_hi = slot + n if self._ln_hi is None else max(self._ln_hi, slot + n)
        return self._ln_arena[0, slot:slot + n]

    def fold_copies(self, early=False):
        """sum the accumulator copies (LayerNorm parameters) and the tap-ordered conv weight gradients touched since the last fold into G.
        early=True marks the folds whose only purpose is to finish a range before it is handed to the data-parallel wrapper (after the MIM decoder, after
        each trunk stage): without a wrapper they wait for the fold at the end of the trunk's backward -- two launches per step instead of nine
        (every small launch between two large ones costs several times its own duration in drained pipelines)."""
        from . import ops
        if early and self.on_range_ready is None and not _EARLY_FOLDS:
            return
        if self.G.is_cuda:
            self.tn_fold_flush()             # deferred partial-tile folds of the weight-gradient GEMMs (their conv outputs land in the tap arena folded next)
        if getattr(self, "_tap_lo", None) is not None:
            ops.fold_copies(self._tap_arena, 1, self._tap_arena.numel(), self._tap_index, self._tap_lo, self._tap_hi, self.G)
            self._tap_lo = self._tap_hi = None
        if getattr(self, "_ln_lo", None) is None:
            return
        ops.fold_copies(self._ln_arena, self.LN_COPIES, self.ln_stride, self._ln_index, self._ln_lo, self._ln_hi, self.G)
        self._ln_lo = self._ln_hi = None

    # ---- conv weight gradients in the gather's order ----------------------------------------------------------------------
    # The gathered TN GEMM produces a conv weight gradient as [out][kh][kw][cin]; nn.Conv2d keeps [out][cin][kh][kw].  Each conv
    # weight owns a slot of a persistent fp32 arena that the GEMM accumulates into, and the fold launch above adds the touched range
    # to G through an index table (arena element (o, t, c) -> G element (o, c, t)) and zeroes it again: one launch per backward
    # stage instead of one permuted ATen add per convolution (24 per step).
    def grad_taps(self, name, cout, taps, cin):
        """arena view [cout][taps * cin] standing in for self.grad(name) as the output of the gathered weight-gradient GEMM"""
        if getattr(self, "_tap_arena", None) is None or self._tap_arena.device != self.G.device:
            total = sum(n for _, n, shape in self.offsets.values() if len(shape) == 4)
            self._tap_arena = torch.zeros((total + 63) // 64 * 64, device=self.G.device, dtype=torch.float32)
            self._tap_index = torch.zeros(self._tap_arena.numel(), device=self.G.device, dtype=torch.int32)
            self._tap_slots, self._tap_next, self._tap_lo, self._tap_hi = {}, 0, None, None
        goff, n, _ = self.offsets[name]
        assert n == cout * taps * cin, (name, n, cout, taps, cin)
        slot = self._tap_slots.get(name)
        if slot is None:
            slot = self._tap_slots[name] = self._tap_next
            dev = self.G.device
            o = torch.arange(cout, device=dev, dtype=torch.int64)[:, None, None]
            t = torch.arange(taps, device=dev, dtype=torch.int64)[None, :, None]
            c = torch.arange(cin, device=dev, dtype=torch.int64)[None, None, :]
            self._tap_index[slot:slot + n] = (goff + (o * cin + c) * taps + t).reshape(-1).to(torch.int32)
            self._tap_next += n
        self._tap_lo = slot if self._tap_lo is None else min(self._tap_lo, slot)
        self._tap_hi = slot + n if self._tap_hi is None else max(self._tap_hi, slot + n)
        return self._tap_arena[slot:slot + n].view(cout, taps * cin)

    def announce_stage(self, i):
        """backward of stage i finished: its parameter gradients are final -> let the data-parallel wrapper start
        reducing them while the earlier stages are still running."""
        self.fold_copies(early=True)
        if self.on_range_ready is not None:
            lo, hi = self.stage_range(i)
            self._ranges_done.append((lo, hi))
            self.on_range_ready(self, lo, hi)

    def prefix_range(self, prefixes):
        """[lo, hi) of the parameters whose names start with one of `prefixes` (they are contiguous in registration order)"""
        names = [n for n in self.offsets if n.startswith(prefixes)]
        if not names:
            return None
        lo = self.offsets[names[0]][0]
        o, n, _ = self.offsets[names[-1]]
        hi = o + (n + ALIGN - 1) // ALIGN * ALIGN
        assert sum((self.offsets[k][1] + ALIGN - 1) // ALIGN * ALIGN for k in names) == hi - lo, ("not contiguous", prefixes)
        return lo, hi

    def announce_prefix(self, *prefixes):
        """a head's backward finished: the gradients of its own parameters are final (heads run before the trunk in a backward
        pass, so these ranges travel while the whole trunk backward is still ahead)"""
        if self.on_range_ready is None:
            return
        r = self.prefix_range(tuple(prefixes))
        if r is not None:
            if self.G.is_cuda:
                self.tn_fold_flush()      # a range leaves only with its deferred folds applied (no-op when nothing is pending: no head defers today -- ADVICE r5)
            self._ranges_done.append(r)
            self.on_range_ready(self, *r)

    def scale_grads(self, factor):
        """G *= factor, now or (when the fused optimizer owns the next step) inside its kernel."""
        if self.scale_in_optimizer:
            self.pending_grad_scale *= factor
        else:
            self.G.mul_(factor)

    def tn_fold_flush(self):
        """fold what this store's weight-gradient launches left pending in its scratch (nothing allocated: nothing pending)"""
        if getattr(self, "_tn_partials", None) is not None:
            from . import ops
            ops.tn_fold_flush(self._tn_partials)

    def tn_partials(self):
        """256 MiB of persistent scratch for the atomic-free reduction of the weight-gradient GEMMs (mvlt_gemm_tn_args.partials): bf16 partial tiles [splits][N1][N2].  The
        stage-4 MLP needs 256 x 65536 elements (32 MiB), the MIM decoder's largest conv (192 -> 192 at 32 x 32: 56 splits x 192 x 1728) 37 MiB; what does not fit takes atomics."""
        t = getattr(self, "_tn_partials", None)
        if t is None or t.device != self.G.device:
            t = self._tn_partials = torch.empty(_TN_SCRATCH_MIB * 8 * 65536, dtype=torch.bfloat16, device=self.G.device)
        return t

    def wait_grads(self):
        """block (the stream, for RCCL; the host, for gloo) until every gradient collective handed over by the data-parallel wrapper is done"""
        works, self.grad_works = self.grad_works, []
        for w, lo, hi, g, t, early in works:
            w.wait()
            if t is not None:
                g.copy_(t)

    def apply_pending_scale(self):
        """make G itself carry the owed factor (anything that READS gradients before the optimizer step: clipping, logging)"""
        self.wait_grads()
        if self.pending_grad_scale != 1.0:
            self.G.mul_(self.pending_grad_scale)
            self.pending_grad_scale = 1.0

    def sync_grads(self):
        """Make G the truth for every parameter (flat-buffer optimizers call this before stepping): gradients that
        autograd produced on its own (the torch-autograd MIM decoder) or that were cloned / averaged elsewhere
        are copied into their slice and .grad is re-pointed at it."""
        for name, p in self.params.items():
            gv = self.grad(name)
            if p.grad is None:
                continue
            if p.grad.data_ptr() != gv.data_ptr():
                gv.copy_(p.grad)
                p.grad = gv
