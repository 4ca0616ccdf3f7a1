"""Which host lines still launch ATen kernels in one training step: a TorchDispatchMode that records, for every ATen op that is
not a pure view, the innermost mvlt_amd / bench frame on the Python stack (forward, custom-Function backwards and optimizer)."""
import os, sys, collections, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mvlt_amd import pvlt
from mvlt_amd.engine import BF16Scaler, train_step
from mvlt_amd.optim import FusedAdamW
from torch.utils._python_dispatch import TorchDispatchMode
dev = torch.device('cuda', 0)
model = pvlt.pvlt_tiny(pretrained=False, token_hidden_size=768, num_text_tokens=128, loss_type=dict(mlm=1, itm=1, t2i=1, cls=0),
                       pretrained_pth=None, drop_path_rate=0.1, drop_rate=0.0, num_classes=1000, in_chans=3).cuda(dev)
model.train()
B = 64
batch = bench.synth_batch(B, 256, 128, dev, 1)
opt = FusedAdamW(model, lr=1e-4, weight_decay=0.01); scaler = BF16Scaler()
import argparse, contextlib
from mvlt_amd import engine as E
eargs = argparse.Namespace(loss_type=dict(mlm=1, itm=1, t2i=1, cls=0))
NSTEP = 4
def step(i):
    with contextlib.redirect_stdout(sys.stderr):
        E.train_one_epoch_vl(model, None, [batch] * NSTEP, opt, dev, i, scaler, None, None, None, True, False, eargs)
for i in range(2): step(i)
torch.cuda.synchronize()
VIEWS = {"view", "as_strided", "reshape", "slice", "select", "permute", "transpose", "t", "_unsafe_view", "expand", "unsqueeze", "squeeze", "detach",
         "alias", "_local_scalar_dense", "narrow", "unflatten", "flatten", "empty", "empty_strided", "empty_like", "new_empty", "new_empty_strided",
         "_reshape_alias", "lift_fresh", "unbind", "split", "split_with_sizes", "chunk", "is_pinned", "record_stream", "set_", "resize_", "is_same_size",
         "sym_size", "sym_numel", "sym_stride", "sym_storage_offset", "stride", "size", "numel", "dim", "view_as_real", "unsafe_split", "_pin_memory"}
cnt = collections.Counter()
class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        out = func(*args, **(kwargs or {}))
        if name not in VIEWS:
            outs = out if isinstance(out, (tuple, list)) else (out,)
            on_dev = any(isinstance(a, torch.Tensor) and a.is_cuda for a in list(args) + list((kwargs or {}).values()) + list(outs))
            fr = [f for f in traceback.extract_stack() if ("mvlt_amd/" in f.filename or f.filename.endswith("bench.py")) and "_lib.py" not in f.filename]
            where = f"{fr[-1].filename.split('mvlt_amd/')[-1]}:{fr[-1].lineno} {fr[-1].name}" if fr else "?"
            shape = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), ())
            if not shape:
                shape = next((tuple(a.shape) for a in outs if isinstance(a, torch.Tensor)), ())
            cnt[(name, where, on_dev, str(shape)[:28])] += 1
        return out
with Rec():
    step(3)
torch.cuda.synchronize()
tot = 0
for (name, where, on_dev, shape), c in sorted(cnt.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    if on_dev:
        tot += c
        print(f"{c:4d}  {name:24s} {shape:28s} {where}")
print("device ATen ops per step:", tot / NSTEP, "(counts above are over", NSTEP, "engine iterations + epoch end)")
