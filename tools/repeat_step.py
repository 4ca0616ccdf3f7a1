"""Run the same pre-train step (same batch, same dropout / DropPath draws) N times and compare every gradient with the first
run: anything beyond fp32-atomic-order noise is a race or an uninitialised read.   python tools/repeat_step.py [B] [N]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mvlt_amd import pvlt
from mvlt_amd.engine import train_step
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
model = pvlt.pvlt_tiny(pretrained=False, token_hidden_size=768, num_text_tokens=128, loss_type=dict(mlm=1, itm=1, t2i=1, cls=0),
                       pretrained_pth=None, drop_path_rate=0.1, drop_rate=0.0, num_classes=1000, in_chans=3).cuda(dev)
model.train()
batch = bench.synth_batch(B, 256, 128, dev, 1)
batch["mlm_positions"] = torch.nonzero(batch["mlm_labels"].reshape(-1) != -1).flatten().to(torch.int32)
ref = None
for it in range(N):
    model.store.G.zero_() if getattr(model, "store", None) is not None and model.store.G is not None else None
    for p in model.parameters():
        if p.grad is not None: p.grad.zero_()
    torch.manual_seed(1234); torch.cuda.manual_seed(1234); model._rng_calls = 0
    junk = [torch.full((1 << 24,), float("nan"), device=dev) for _ in range(6)]; del junk       # poison freed memory
    total, _ = train_step(model, batch, 1, True)
    total.backward()
    torch.cuda.synchronize()
    cur = {k: p.grad.detach().float().clone() for k, p in model.named_parameters() if p.grad is not None}
    if ref is None:
        ref = cur; print("loss", total.item(), "params with grad", len(cur), flush=True); continue
    worst = sorted(((cur[k] - ref[k]).norm().item() / max(ref[k].norm().item(), 1e-20), k) for k in ref)[-3:]
    print("iter", it, "loss", round(total.item(), 6), "worst rel grad deviation", [(f"{d:.2e}", k) for d, k in reversed(worst)], flush=True)
