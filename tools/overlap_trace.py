#!/usr/bin/env python3
"""Where the gradient all-reduces are issued against the backward kernels (VERDICT r3 #6a), on a ONE-GPU box.

A rocprofv3 kernel trace of the rank process at forced world size 1 (launcher variables set by hand, the program directly behind `--`)
shows NO RCCL kernel: with one rank ProcessGroupNCCL / RCCL complete an all-reduce without launching anything (tried, round 4: 420 kernels
per step on one queue, none of them RCCL's), and two ranks cannot share one GPU under RCCL.  What CAN be shown here is the point of the
backward pass at which each collective becomes eligible to run: ProcessGroupNCCL makes its stream wait for an event recorded on the
compute stream at the moment `all_reduce(..., async_op=True)` is called, i.e. behind the last kernel that wrote the range.  This tool
records such an event at every `DataParallel._reduce` call of a real engine iteration (batch 256, forced collectives at world size 1) and
one behind the last backward kernel, and prints when each range is released relative to the start and the end of the backward pass --
everything released before the end can overlap the remaining backward kernels, what is released at the end cannot.  Link time and the
actual concurrency at N > 1 are UNMEASURED ON HARDWARE.

    RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 MVLT_DP_FORCE_COLLECTIVES=1 python tools/overlap_trace.py > profiles/r04_overlap.txt
"""
import argparse
import contextlib
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    assert os.environ.get("MVLT_DP_FORCE_COLLECTIVES") and "WORLD_SIZE" in os.environ, __doc__
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=dev)
    from mvlt_amd import pvlt
    from mvlt_amd.dist import DataParallel
    from mvlt_amd.engine import BF16Scaler
    from mvlt_amd.optim import FusedAdamW
    import engine_grid_masking as E
    lt = dict(mlm=1, itm=1, t2i=1, cls=0)
    core = pvlt.pvlt_tiny(pretrained=False, token_hidden_size=768, num_text_tokens=128, loss_type=lt, pretrained_pth=None, drop_path_rate=0.1).cuda(dev)
    model = DataParallel(core)
    batch = bench.synth_batch(256, 256, 128, dev, 1)
    opt = FusedAdamW(core, lr=1e-4, weight_decay=0.01)
    scaler = BF16Scaler()
    eargs = argparse.Namespace(loss_type=lt)

    def epoch(n, ep):
        with contextlib.redirect_stdout(sys.stderr):
            E.train_one_epoch_vl(model, None, [batch] * n, opt, dev, ep, scaler, None, None, None, True, False, eargs)

    epoch(3, 0)
    torch.cuda.synchronize()
    log = []
    S = core.store
    inner_reduce, inner_begin, inner_finish = model._reduce, S.begin_backward, model._finish

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def reduce(store, lo, hi):
        log.append(("reduce", ev(), lo, hi))
        inner_reduce(store, lo, hi)

    def begin():
        log.append(("begin", ev(), 0, 0))
        inner_begin()

    def finish(store):
        log.append(("bwd_end", ev(), 0, 0))
        inner_finish(store)
        log.append(("finish", ev(), 0, 0))

    model._reduce, S.begin_backward, S.on_backward_done = reduce, begin, finish
    epoch(2, 1)
    torch.cuda.synchronize()
    # last iteration
    last = max(i for i, r in enumerate(log) if r[0] == "begin")
    it = log[last:]
    t0 = it[0][1]
    t_end = next(r[1] for r in it if r[0] == "bwd_end")
    bwd_ms = t0.elapsed_time(t_end)
    names = sorted(S.offsets.items(), key=lambda kv: kv[1][0])

    def what(lo, hi):
        inside = [n for n, (o, k, _) in names if lo <= o < hi]
        heads = sorted({n.split(".")[0] for n in inside})
        return ", ".join(heads[:6]) + (" ..." if len(heads) > 6 else "")

    print(f"pvlt_tiny pre-train, batch 256, one MI355X, collectives forced at world size 1 (RCCL launches no kernel for them: see the docstring)")
    print(f"backward pass: {bwd_ms:.3f} ms of GPU time from its first HIP-scheduled node to its last kernel")
    print(f"{'released at':>12s} {'% of bwd':>9s} {'MB':>8s}  parameters in the range")
    early = 0.0
    total = 0.0
    for kind, e, lo, hi in it:
        if kind != "reduce":
            continue
        t = t0.elapsed_time(e)
        mb = (hi - lo) * 4 / 2 ** 20
        total += mb
        if t < bwd_ms - 0.05:
            early += mb
        print(f"{t:9.3f} ms {100.0 * t / bwd_ms:8.1f}% {mb:8.2f}  {what(lo, hi)}")
    print(f"{early:.1f} of {total:.1f} MB ({100.0 * early / total:.0f} %) are released before the backward pass ends; at the assumed ~150 GB/s per ring direction "
          f"(unmeasured) the {total - early:.1f} MB released at the end cost ~{(total - early) / 150e3 * 2 * 7 / 8 * 1e3:.2f} ms of un-overlapped all-reduce at N = 8")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
