#!/usr/bin/env python3
"""Headline benchmark: image-text pairs / second / node of the PVT-tiny MVLT pre-train step (MLM + MIM + ITM;
forward + loss + backward + gradient all-reduce + AdamW) on N MI355X of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: pvlt_tiny, 256x256 RGB + 128 BERT tokens, batch 256 per GPU, bf16 MFMA
operands with fp32 accumulation / residual stream / master weights, synthetic data generated on device
(SURVEY.md 8d), weak scaling (per-GPU batch fixed).  A step is one engine iteration (reference
engine_grid_masking.py:40-143): clean image on even steps, grid-masked image on odd steps.

Prints ONE JSON line (rank 0) with the driver's contract plus
  roofline      the dominant kernel timed live with HIP events on its launch stream (see DESIGN.md for the
                algorithmic FLOP count used)
  cpu_baseline  the CPU oracle (oracle/pvlt_oracle.py, kind "port") timed on this box's host cores at config #1
                shapes (4 pairs), N=1 only
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_PAIR_TRAIN = 50.38e9      # 3 x 16.793 GFLOP forward (reference-equivalent work)
PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def synth_batch(B, S, T, device, seed):
    """Synthetic pairs with the engine's batch schema, generated on the device."""
    g = torch.Generator(device=device).manual_seed(seed)
    image = torch.rand(B, 3, S, S, device=device, generator=g)
    gp = S // 16
    order = torch.rand(B, gp * gp, device=device, generator=g).argsort(dim=1)
    pm = torch.zeros(B, gp * gp, device=device, dtype=torch.bool)
    pm.scatter_(1, order[:, : gp * gp // 2], True)                       # exactly half of the 16x16 patches
    pm = pm.view(B, 1, gp, gp).repeat_interleave(16, 2).repeat_interleave(16, 3)
    masked = torch.where(pm, torch.full_like(image, 1e-6), image)
    L = torch.randint(20, 61, (B,), device=device, generator=g)
    pos = torch.arange(T, device=device)[None]
    tok = torch.randint(1000, 30522, (B, T), device=device, generator=g)
    ori = torch.where(pos == 0, torch.full_like(tok, 101), tok)
    ori = torch.where(pos == (L[:, None] + 1), torch.full_like(tok, 102), ori)
    ori = torch.where(pos > (L[:, None] + 1), torch.zeros_like(tok), ori)
    cap = (pos >= 1) & (pos <= L[:, None])
    sel = cap & (torch.rand(B, T, device=device, generator=g) < 0.15)
    sel[:, 1] |= ~sel.any(dim=1)                                          # at least one selected token per caption
    how = torch.rand(B, T, device=device, generator=g)
    ids = torch.where(sel & (how < 0.8), torch.full_like(tok, 103), ori)
    ids = torch.where(sel & (how >= 0.8) & (how < 0.9), torch.randint(1000, 30522, (B, T), device=device, generator=g), ids)
    labels = torch.where(sel, ori, torch.full_like(ori, -1))
    return dict(image=image, masked_images=masked, input_ids=ids, ori_input_ids=ori, mlm_labels=labels, i2t_labels=ori.clone(),
                itm_labels=torch.randint(0, 2, (B, 1), device=device, generator=g),
                sup_cls_labels=torch.randint(0, 48, (B, 1), device=device, generator=g),
                sub_cls_labels=torch.randint(0, 122, (B, 1), device=device, generator=g))


def cpu_baseline(seconds=20.0):
    """The oracle's train step (forward + loss + backward) at BASELINE config #1 shapes on the host cores."""
    from oracle import filler
    from oracle import pvlt_oracle as O
    from oracle.hostinfo import usable_cores
    cores = usable_cores()
    torch.set_num_threads(cores)
    cfg = O.Cfg("pvlt_tiny", dict(mlm=1, itm=1, t2i=1, cls=0), 224, 768, 128, 0.0)
    sd = O.filled_state_dict(cfg, 7)
    sdg = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and "running_" not in k) else v) for k, v in sd.items() if k != O.TIED[0]}
    sdg[O.TIED[0]] = sdg[O.TIED[1]]
    batch = O.to_torch_batch(filler.make_batch(7, 4, 256, 128))
    n, t0 = 0, None
    while True:
        ls, _ = O.step_loss(sdg, cfg, batch, n, train=True, masks=None, bn_out={})
        ls["total_loss"].backward()
        for v in sdg.values():
            if v.is_floating_point() and v.grad is not None:
                v.grad = None
        n += 1
        if t0 is None:                    # first iteration = warm-up
            t0, n = time.time(), 0
        elif time.time() - t0 > seconds and n >= 3:
            break
    dt = time.time() - t0
    return dict(value=round(4 * n / dt, 3), unit="pairs/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{n} train steps (fwd+loss+bwd, fp32) of 4 pairs 256x256+128 tok on the CPU oracle, {dt:.1f} s")


def _time_launch(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()                    # torch's current stream == the stream the C ABI launches on
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def roofline_cases(B, device):
    """(name, launch closure, algorithmic work) of the two roofline launches; also used by tools/roofline_launch.py."""
    from mvlt_amd import ops
    from mvlt_amd._lib import conv3map
    bf = torch.bfloat16
    M, C = B * 32 * 32, 192
    x = torch.randn(M, C, device=device).to(bf)
    w = (torch.randn(C, 9 * C, device=device) * (9 * C) ** -0.5).to(bf)
    out = torch.empty(M, C, device=device, dtype=torch.float32)
    amap = conv3map(32, 32, 32 * 32, C)
    M2 = B * 4224
    x2 = torch.randn(M2, 64, device=device).to(bf)
    w2 = (torch.randn(64, 64, device=device) * 0.125).to(bf)
    b2 = torch.randn(64, device=device)
    o2 = torch.empty(M2, 64, device=device, dtype=bf)
    return [("conv192", lambda: ops.gemm_nt(x, w, out, M, C, 9 * C, C, 9 * C, C, a_map=amap), 2.0 * M * C * 9 * C),
            ("proj64", lambda: ops.gemm_nt(x2, w2, o2, M2, 64, 64, 64, 64, 64, bias=b2), 2.0 * (2 * M2 * 64 + 64 * 64))]


def time_dominant_kernel(model, B, device):
    """Roofline of the dominant kernel family of the step, `gemm_nt_dma_kernel` (~30 % of GPU time in profiles/): its
    largest in-step launch is the MIM decoder's 192->192 conv3x3 at 32x32 (conv4 / conv_concat3 forward and their
    input gradients) = a gathered-row GEMM with M = B*1024, N = 192, K = 9*192, MFMA-bound (AI ~ 575 F/B).  Timed
    with HIP events on torch's current stream, which is the stream mvlt_gemm_nt launches on.  `traffic` = HBM bytes
    per launch from the committed PMC passes (profiles/*_roofline_traffic.json: FETCH_SIZE x2 + WRITE_SIZE, collected
    offline because counters cannot be read inside this process).  Also reported: the HBM-bound K=64 shape
    (stage-1 q/proj-like projection) of the same kernel family."""
    (_, f1, flops), (_, f2, bytes2) = roofline_cases(B, device)
    ms, ms2 = _time_launch(f1), _time_launch(f2)
    tf = flops / (ms * 1e-3) / 1e12
    traffic = traffic2 = None
    tj = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.endswith("_roofline_traffic.json")) if B == 256 else []
    if tj:
        t = json.load(open(os.path.join(ROOT, "profiles", tj[-1])))
        traffic, traffic2 = t.get("conv192", {}).get("hbm_bytes"), t.get("proj64", {}).get("hbm_bytes")
    return dict(kernel="gemm_nt_dma_kernel<192, 2, 1, 64, 128> (bf16, 128x192 tile, 3x3-gather A, plain epilogue): MIM conv3x3 192->192 @32x32 as GEMM (M=B*1024, N=192, K=1728)",
                bound="mfma", achieved=round(tf, 1), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=round(tf / PEAK_BF16_TFLOPS, 4),
                traffic=traffic, ms_per_launch=round(ms, 4), algorithmic_flops=flops,
                hbm_bound_sibling=dict(kernel="gemm_nt_dma_kernel<64, 0, 1, 64, 128> (bf16, 128x64 tile): K=64 N=64 projection, M=B*4224", bound="hbm",
                                       achieved=round(bytes2 / (ms2 * 1e-3) / 1e9, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                                       frac=round(bytes2 / (ms2 * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), ms_per_launch=round(ms2, 4),
                                       algorithmic_bytes=bytes2, traffic=traffic2))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="pairs per GPU")
    ap.add_argument("--img", type=int, default=256)
    ap.add_argument("--model", default="pvlt_tiny")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from mvlt_amd import pvlt
    from mvlt_amd.dist import DataParallel
    from mvlt_amd.engine import BF16Scaler, train_step
    from mvlt_amd.optim import FusedAdamW

    torch.manual_seed(1234 + rank)
    loss_type = dict(mlm=1, itm=1, t2i=1, cls=0)
    model = getattr(pvlt, args.model)(pretrained=False, token_hidden_size=768, num_text_tokens=128, loss_type=loss_type,
                                      pretrained_pth=None, drop_path_rate=0.1, drop_rate=0.0, num_classes=1000, in_chans=3)
    model.cuda(device)
    core = model
    if world > 1:
        model = DataParallel(model)
    model.train()
    B = args.batch
    batch = synth_batch(B, args.img, 128, device, 1234 + rank)
    batch["mlm_positions"] = torch.nonzero(batch["mlm_labels"].reshape(-1) != -1).flatten().to(torch.int32)
    lr = 2.5e-4 * B * world / 512.0                       # reference main_vl.py:306
    with torch.no_grad():                                 # build the flat store before the optimizer looks at it
        core.eval()
        core(batch["image"][:2], batch["input_ids"][:2])
        core.train()
    opt = FusedAdamW(core, lr=lr, weight_decay=0.01)
    scaler = BF16Scaler()

    def step(i):
        total, parts = train_step(model, batch, i, True)
        opt.zero_grad()
        scaler(total, opt, clip_grad=None, parameters=None)
        return total

    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(args.steps):
        last = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.time() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss_val = float(last)
    pairs_s = B * world * args.steps / dt

    if rank == 0:
        line = {
            "metric": "image-text pairs/sec/node, PVT-tiny MVLT pre-train step", "value": round(pairs_s, 2), "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.model} MVLT pre-train (MLM+MIM+ITM), {args.img}x{args.img} RGB + 128 tokens, "
                                   f"batch {B}/GPU (global {B * world}), fwd+loss+bwd+allreduce+AdamW",
                       "global_batch": B * world, "parallelism": f"dp{world}", "optimizer": "fused AdamW (fp32 master)",
                       "final_loss": round(loss_val, 4)},
            "step_tflops_reference_equivalent": round(pairs_s * FLOP_PER_PAIR_TRAIN / 1e12, 1),
            "mfma_frac_reference_equivalent": round(pairs_s * FLOP_PER_PAIR_TRAIN / 1e12 / (PEAK_BF16_TFLOPS * world), 4),
        }
        if args.model == "pvlt_tiny":
            line["roofline"] = time_dominant_kernel(core, B, device)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
