"""Inference forward (model.eval(), no_grad) of the pre-train model at batch 256: with the MIM decoder's BatchNorms folded into their convs
(default) and with the separate normalisation pass (MVLT_MIM_NO_BN_FOLD semantics, toggled in-process)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mvlt_amd import pvlt, mim
dev = torch.device('cuda', 0)
model = pvlt.pvlt_tiny(pretrained=False, token_hidden_size=768, num_text_tokens=128, loss_type=dict(mlm=1, itm=1, t2i=1, cls=0),
                       pretrained_pth=None, drop_path_rate=0.1, drop_rate=0.0, num_classes=1000, in_chans=3).cuda(dev).eval()
b = bench.synth_batch(256, 256, 128, dev, 1)
def run(n):
    with torch.no_grad():
        for _ in range(n): model(b["image"], b["input_ids"], mlm_labels=b["mlm_labels"])
for nofold in (False, True, False, True):
    mim._NO_BN_FOLD = nofold
    run(3); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(10); e1.record(); torch.cuda.synchronize()
    print("eval forward, batch 256, %s: %.2f ms" % ("separate BatchNorm pass" if nofold else "BatchNorm folded", e0.elapsed_time(e1) / 10), flush=True)
