"""Which host lines still launch ATen kernels in one training step (torch.profiler with stacks)."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mvlt_amd import pvlt
from mvlt_amd.engine import BF16Scaler, train_step
from mvlt_amd.optim import FusedAdamW
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda', 0)
model = pvlt.pvlt_tiny(pretrained=False, token_hidden_size=768, num_text_tokens=128, loss_type=dict(mlm=1, itm=1, t2i=1, cls=0),
                       pretrained_pth=None, drop_path_rate=0.1, drop_rate=0.0, num_classes=1000, in_chans=3).cuda(dev)
model.train()
B = 64
batch = bench.synth_batch(B, 256, 128, dev, 1)
batch["mlm_positions"] = torch.nonzero(batch["mlm_labels"].reshape(-1) != -1).flatten().to(torch.int32)
with torch.no_grad():
    model.eval(); model(batch["image"][:2], batch["input_ids"][:2]); model.train()
opt = FusedAdamW(model, lr=1e-4, weight_decay=0.01); scaler = BF16Scaler()
def step(i):
    total, _ = train_step(model, batch, i, True)
    opt.zero_grad(); scaler(total, opt, clip_grad=None, parameters=None)
for i in range(3): step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    step(3)
torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.name in ("aten::empty", "aten::empty_strided", "aten::view", "aten::as_strided", "aten::reshape",
                                                           "aten::slice", "aten::select", "aten::permute", "aten::transpose", "aten::t", "aten::_unsafe_view",
                                                           "aten::expand", "aten::unsqueeze", "aten::squeeze", "aten::detach", "aten::alias", "aten::item",
                                                           "aten::_local_scalar_dense", "aten::empty_like", "aten::narrow", "aten::unflatten", "aten::flatten", "aten::contiguous",
                                                           "aten::to", "aten::_to_copy", "aten::clone", "aten::zeros", "aten::zeros_like", "aten::result_type", "aten::lift_fresh", "aten::resolve_conj", "aten::resolve_neg"):
        continue
    st = [f for f in (ev.stack or []) if ("mvlt_amd/" in f or "bench.py" in f) and "_lib.py" not in f]
    where = st[0].split("mvlt_amd/")[-1] if st else (ev.stack[0] if ev.stack else "?")
    cnt[(ev.name, where[:90])] += 1
for (name, where), c in cnt.most_common(60):
    print(f"{c:4d}  {name:28s} {where}")
