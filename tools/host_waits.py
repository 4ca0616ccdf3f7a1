"""Where the host waits inside train_one_epoch_vl: time blocked in the loss read-back (waits for the iteration's forward: the host is
AHEAD when this is large) and in the masked-row count of the MLM head (waits for a copy queued at the head of the forward: large when the
queue in front of it is long), per step, for the pre-train and the fine-tune step.  python tools/host_waits.py [steps]"""
import argparse, contextlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mvlt_amd import pvlt, engine, schedule
from mvlt_amd.optim import FusedAdamW
dev = torch.device('cuda', 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
waits = dict(readback=0.0, count=0.0, n_count=0)
_rb, _hc = engine._LossReadback.get, schedule._HostCount.get
def rb(self):
    t = time.perf_counter(); r = _rb(self); waits["readback"] += time.perf_counter() - t; return r
def hc(self):
    t = time.perf_counter(); r = _hc(self); waits["count"] += time.perf_counter() - t; waits["n_count"] += 1; return r
engine._LossReadback.get, schedule._HostCount.get = rb, hc
for task, lt in (("pretrain", dict(mlm=1, itm=1, t2i=1, cls=0)), ("finetune", dict(mlm=0, itm=0, t2i=0, cls=1))):
    torch.manual_seed(1)
    model = pvlt.pvlt_tiny(pretrained=False, token_hidden_size=768, num_text_tokens=128, loss_type=lt, pretrained_pth=None, drop_path_rate=0.1,
                           drop_rate=0.0, num_classes=1000, in_chans=3).cuda(dev)
    batch = bench.synth_batch(256, 256, 128, dev, 1)
    opt = FusedAdamW(model, lr=1e-4, weight_decay=0.01); scaler = engine.BF16Scaler()
    eargs = argparse.Namespace(loss_type=lt)
    def epoch(n, ep):
        with contextlib.redirect_stdout(sys.stderr):
            engine.train_one_epoch_vl(model, None, [batch] * n, opt, dev, ep, scaler, None, None, None, True, False, eargs)
    epoch(5, 0); torch.cuda.synchronize()
    for k in waits: waits[k] = 0
    t0 = time.perf_counter(); epoch(steps, 1); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{task}: {1e3 * dt / steps:.2f} ms/step; host blocked in loss read-back {1e3 * waits['readback'] / steps:.2f} ms/step, in the MLM count "
          f"{1e3 * waits['count'] / steps:.2f} ms/step ({waits['n_count'] / steps:.1f} waits/step); host busy {1e3 * (dt - waits['readback'] - waits['count']) / steps:.2f} ms/step", flush=True)
    del model, opt
