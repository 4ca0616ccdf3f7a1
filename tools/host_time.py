"""Host time to enqueue one training step (forward + backward + optimizer, no read-back) against the GPU time of the step: how far the
host is from becoming the bottleneck."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mvlt_amd import pvlt
from mvlt_amd.engine import BF16Scaler, train_step
from mvlt_amd.optim import FusedAdamW
dev = torch.device('cuda', 0)
model = getattr(pvlt, os.environ.get("MODEL", "pvlt_tiny"))(pretrained=False, token_hidden_size=768, num_text_tokens=128, loss_type=dict(mlm=1, itm=1, t2i=1, cls=0),
                       pretrained_pth=None, drop_path_rate=0.1, drop_rate=0.0, num_classes=1000, in_chans=3).cuda(dev)
model.train()
B, IMG = int(os.environ.get("B", "256")), int(os.environ.get("IMG", "256"))        # MODEL=pvlt_medium IMG=384 B=64: BASELINE configuration #4
batch = bench.synth_batch(B, IMG, 128, dev, 1)
batch["mlm_count"] = int((batch["mlm_labels"] != -1).sum())
opt = FusedAdamW(model, lr=1e-4, weight_decay=0.01); scaler = BF16Scaler()
def step(i):
    total, _ = train_step(model, batch, i, True)
    opt.zero_grad(); scaler(total, opt, clip_grad=None, parameters=None)
for i in range(3): step(i)
torch.cuda.synchronize()
N = 8
t0 = time.perf_counter()
for i in range(N): step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / N:.2f} ms/step; GPU-bound total {1e3 * (t2 - t0) / N:.2f} ms/step")
# what one launch costs the host: 2000 tiny launches through the C ABI (a 4-element cast), queue never drained
from mvlt_amd import ops
src, dst = torch.zeros(4, device=dev), torch.zeros(4, device=dev, dtype=torch.bfloat16)
for _ in range(100): ops.cast_bf16(src, dst, 4)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): ops.cast_bf16(src, dst, 4)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host cost of one launch through ops.* (ctypes + hipLaunchKernel, tiny kernel): {1e6 * (t1 - t0) / 2000:.1f} us; x 377 launches = {377 * 1e3 * (t1 - t0) / 2000:.2f} ms/step")
