#!/usr/bin/env python3
"""Per-shape GEMM inventory of one pre-train step: records every ops.gemm_nt / ops.gemm_tn / fused-MLP call of one
engine iteration, then re-times each distinct shape alone (HIP events, 10 reps) and prints count x time per step.

    gpurun -- 'python tools/gemm_shapes.py > gpurun_out/gemm_shapes.txt'
    MODEL=pvlt_medium IMG=384 B=64 python tools/gemm_shapes.py        (BASELINE configuration #4)
"""
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mvlt_amd import ops, pvlt  # noqa: E402
from mvlt_amd.engine import BF16Scaler, train_step  # noqa: E402
from mvlt_amd.optim import FusedAdamW  # noqa: E402


def main():
    B = int(os.environ.get("B", "256"))
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    IMG = int(os.environ.get("IMG", "256"))
    model = getattr(pvlt, os.environ.get("MODEL", "pvlt_tiny"))(pretrained=False, token_hidden_size=768, num_text_tokens=128, loss_type=dict(mlm=1, itm=1, t2i=1, cls=0),
                           pretrained_pth=None, drop_path_rate=0.1, drop_rate=0.0, num_classes=1000, in_chans=3).cuda(dev)
    model.train()
    batch = bench.synth_batch(B, IMG, 128, dev, 1)
    batch["mlm_positions"] = torch.nonzero(batch["mlm_labels"].reshape(-1) != -1).flatten().to(torch.int32)
    with torch.no_grad():
        model.eval(); model(batch["image"][:2], batch["input_ids"][:2]); model.train()
    opt = FusedAdamW(model, lr=1e-4, weight_decay=0.01)
    scaler = BF16Scaler()

    def step(i):
        total, _ = train_step(model, batch, i, True)
        opt.zero_grad()
        scaler(total, opt, clip_grad=None, parameters=None)

    for i in range(3):
        step(i)
    torch.cuda.synchronize()

    rec = OrderedDict()
    orig = {}

    def wrap(name, keyfn):
        f = getattr(ops, name)
        orig[name] = f

        def g(*a, **k):
            key = (name,) + keyfn(a, k)
            if key not in rec:
                rec[key] = [0, a, k]
            rec[key][0] += 1
            return f(*a, **k)
        setattr(ops, name, g)

    def mp(m):
        return "-" if m is None or m.mode == 0 and m.rows_per_batch == 0 else f"m{m.mode}r{m.r}"

    wrap("gemm_nt", lambda a, k: (a[3], a[4], a[5], "A:" + mp(k.get("a_map")), "C:" + mp(k.get("c_map")),
                                  "b" if k.get("bias") is not None else "", f"act{k.get('act', 0)}",
                                  "R" if k.get("R") is not None else "", str(a[2].dtype)[6:]))
    wrap("gemm_tn", lambda a, k: tuple(x for x in a if isinstance(x, int))[:3] + ("A:" + mp(k.get("a_map")), "B:" + mp(k.get("b_map")),
                                                                                   "cs" if k.get("colsum") is not None else ""))
    for nm in ("mlp_fwd", "mlp_bwd_dx", "mlp_bwd_dw"):
        if hasattr(ops, nm):
            wrap(nm, lambda a, k: tuple(x for x in a if isinstance(x, int))[:3])
    # SR attention: key = (B, heads, queries, keys)
    for nm in ("sr_attention_fwd", "sr_attention_bwd"):
        if hasattr(ops, nm):
            wrap(nm, lambda a, k: tuple(x for x in a if isinstance(x, int))[:4])
    step(3)
    torch.cuda.synchronize()
    for nm, f in orig.items():
        setattr(ops, nm, f)

    rows = []
    for key, (cnt, a, k) in rec.items():
        f = orig[key[0]]
        for _ in range(2):
            f(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f(*a, **k)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100.0
        fl = None
        if key[0] in ("gemm_nt", "gemm_tn"):
            fl = 2.0 * key[1] * key[2] * key[3]
        rows.append((cnt * us, cnt, us, fl, key))
    rows.sort(key=lambda r: -r[0])
    tot = sum(r[0] for r in rows)
    print(f"total {tot / 1e3:.2f} ms/step over {sum(r[1] for r in rows)} calls")
    for t, cnt, us, fl, key in rows:
        tf = f"{fl / us / 1e6:7.1f} TF/s" if fl else " " * 12
        print(f"{t / 1e3:7.3f} ms  x{cnt:3d}  {us:8.1f} us  {tf}  {' '.join(str(x) for x in key)}")


if __name__ == "__main__":
    main()
