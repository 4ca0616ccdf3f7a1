import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvlt_amd import ops
dev = torch.device('cuda:0')
for dt in (torch.bfloat16, torch.float32):
    B, S, k = 256, 256, 4
    img = torch.rand(B, 3, S, S, device=dev)
    P = torch.empty(B * (S // k) ** 2, 48, device=dev, dtype=dt)
    ops.patchify(img, P, B, 3, S, S, k)
    ref = img.view(B, 3, S // k, k, S // k, k).permute(0, 2, 4, 1, 3, 5).reshape(-1, 48).to(dt)
    print(dt, "equal:", torch.equal(P, ref))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): ops.patchify(img, P, B, 3, S, S, k)
    e0.record()
    for _ in range(20): ops.patchify(img, P, B, 3, S, S, k)
    e1.record(); torch.cuda.synchronize()
    print(dt, "%.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
