import os, sys, collections, traceback, torch
sys.path.insert(0, '/root/repo')
import bench
from mvlt_amd import pvlt, params
from mvlt_amd.engine import BF16Scaler, train_step
from mvlt_amd.optim import FusedAdamW
dev = torch.device('cuda', 0)
model = pvlt.pvlt_tiny(pretrained=False, token_hidden_size=768, num_text_tokens=128, loss_type=dict(mlm=1, itm=1, t2i=1, cls=0),
                       pretrained_pth=None, drop_path_rate=0.1, drop_rate=0.0, num_classes=1000, in_chans=3).cuda(dev)
model.train()
B = 256
batch = bench.synth_batch(B, 256, 128, dev, 1)
opt = FusedAdamW(model, lr=1e-4, weight_decay=0.01); scaler = BF16Scaler()
def step(i):
    total, _ = train_step(model, batch, i, True)
    opt.zero_grad(); scaler(total, opt, clip_grad=None, parameters=None)
for i in range(2): step(i)
cnt = collections.Counter()
orig = params.ZeroPool.take
def take(self, shape, dtype):
    fr = [f for f in traceback.extract_stack() if 'mvlt_amd/' in f.filename and 'params.py' not in f.filename]
    n = 1
    for d in shape: n *= int(d)
    cnt[(f"{fr[-1].filename.split('mvlt_amd/')[-1]}:{fr[-1].lineno}", tuple(shape), str(dtype))] += n * torch.empty((), dtype=dtype).element_size()
    return orig(self, shape, dtype)
params.ZeroPool.take = take
step(3)
torch.cuda.synchronize()
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:25]:
    print(f"{v/1e6:9.1f} MB  {k}")
print("total MB", sum(cnt.values())/1e6)
