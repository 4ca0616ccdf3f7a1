#!/usr/bin/env python3
"""Headline benchmark: image-text pairs / second / node of the PVT-tiny MVLT pre-train step (MLM + MIM + ITM;
forward + loss + backward + gradient all-reduce + AdamW) on N MI355X of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N --steps K --warmup W          # starts its own N ranks (child processes, before any GPU call)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             # or under a launcher (RANK / LOCAL_RANK / WORLD_SIZE from the environment)

Workload = BASELINE.json configs[1]: pvlt_tiny, 256x256 RGB + 128 BERT tokens, batch 256 per GPU, bf16 MFMA
operands with fp32 accumulation / residual stream / master weights, synthetic data generated on device
(SURVEY.md 8d), weak scaling (per-GPU batch fixed).  A step is one engine iteration (reference
engine_grid_masking.py:40-143): clean image on even steps, grid-masked image on odd steps.

Prints ONE JSON line (rank 0) with the driver's contract plus
  roofline      the largest single launch of the step (fused-MLP weight gradients of a stage-1 block) timed live with HIP events on
                its launch stream: algorithmic vs executed FLOPs, PMC traffic; round 2's MFMA- and HBM-bound launches as `siblings`
  step          executed TFLOP/s and HBM GB per step (committed whole-step counter passes) over ms per step
  cpu_baseline  the CPU oracle (oracle/pvlt_oracle.py, kind "port") timed on this box's host cores at config #1
                shapes (4 pairs), N=1 only: full train step (fwd+loss+bwd+AdamW) = `value`, forward+loss beside it
  flops         reference-equivalent and EXECUTED FLOPs per pair, and the blocks-only (SRAttention + MLP) MFMA utilisation
                north_star asks for: 3 x 8.003 GFLOP/pair over the GPU time between HIP events around the Block kernels

The timed region is `engine_grid_masking.train_one_epoch_vl` itself (the reference's entry point, main_vl.py:431-437) over
a loader of K batches that are already resident in HBM: masked-index selection, loss read-back, zero_grad, backward,
gradient exchange, AdamW and the epoch-end meter reduction are all inside it.
"""
import argparse
import contextlib
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_PAIR_TRAIN = 50.38e9      # 3 x 16.793 GFLOP forward (reference-equivalent work)
# SURVEY.md 8d, forward GFLOP per pair of pvlt_tiny @256/T128: blocks 8.003 (of which MLP 5.184: stage 1-4 blocks
# 0.554/0.604/0.629/0.805 each), MLM head 6.253, everything else 16.793 - 6.253
FWD_BLOCKS, FWD_MLM, FWD_ALL = 8.003e9, 6.253e9, 16.793e9
FWD_FC1_RECOMPUTED = 2 * (0.554e9 + 0.604e9) / 2      # fc1 of the 2+2 fused-MLP blocks of stages 1-2 (half of each block's MLP FLOPs)
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
    """The oracle at BASELINE config #1 shapes (4 pairs, 256x256 + 128 tokens, fp32) on the host cores: forward+loss, and the
    full train step the way the reference's loop runs it (forward, loss, zero_grad, backward, torch.optim.AdamW with timm's
    parameter split -- engine_grid_masking.py:69-127, main_vl.py:308).  BASELINE.md section 5 asks for both."""
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
    opt = torch.optim.AdamW(O.adamw_param_groups([(k, v) for k, v in sdg.items() if k != O.TIED[0] and v.requires_grad], 0.01), lr=1e-5)

    def timed(fn, budget, min_n):
        fn(0)                             # warm-up
        n, t0 = 0, time.time()
        while time.time() - t0 < budget or n < min_n:
            fn(n)
            n += 1
        return n, time.time() - t0

    def fwd(i):
        with torch.no_grad():
            O.step_loss(sdg, cfg, batch, i, train=True, masks=None, bn_out={})

    def step(i):
        ls, _ = O.step_loss(sdg, cfg, batch, i, train=True, masks=None, bn_out={})
        opt.zero_grad()
        ls["total_loss"].backward()
        opt.step()

    nf, tf = timed(fwd, seconds * 0.3, 3)
    ns, ts = timed(step, seconds * 0.7, 3)
    return dict(value=round(4 * ns / ts, 3), unit="pairs/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{ns} train steps (fwd+loss+bwd+AdamW, fp32) of 4 pairs 256x256+128 tok on the CPU oracle in {ts:.1f} s; "
                       f"forward+loss alone: {nf} passes in {tf:.1f} s",
                forward_loss_value=round(4 * nf / tf, 3))


def _time_launch(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()                    # torch's current stream == the stream the C ABI launches on
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    from mvlt_amd._lib import last_kernel
    return e0.elapsed_time(e1) / reps, last_kernel()     # the instantiation the library launched for this call (C ABI: mvlt_last_kernel)


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
    # the roofline launch since round 3 (then the single largest of the step; profiles/r04_gemm_shapes.txt): the fused-MLP weight gradients of a stage-1 block,
    # dW1 / db1 / dW2 / db2 from x and dy with the hidden activation recomputed on chip (M = B*4224 tokens, C = 64, hidden 512), with the
    # per-sample DropPath factors of the step (one sample in ten dropped)
    hid = 512
    xm = torch.randn(M2, 64, device=device).to(bf)
    dym = torch.randn(M2, 64, device=device).to(bf)
    w1 = (torch.randn(hid, 64, device=device) * 0.125).to(bf)
    w2t = (torch.randn(hid, 64, device=device) * hid ** -0.5).to(bf)
    b1 = torch.randn(hid, device=device) * 0.1
    dw1, db1, dw2, db2 = (torch.zeros(hid, 64, device=device), torch.zeros(hid, device=device), torch.zeros(64, hid, device=device),
                          torch.zeros(64, device=device))
    # round 6: the step's OWN factors -- the second block of stage 1 has DropPath rate 0.1 / 7 (the first one 0): keep = 1 / (1 - 0.0143), and
    # ceil(256 x 0.0143) = 4 of 256 samples dropped (the kernel skips their tiles; `algorithmic_flops` counts the kept samples only).  Round 5 dropped one
    # sample in ten here while counting all of them: the headline `frac` erred upward by that share.
    rate = 0.1 / 7.0
    n_drop = max(1, int(round(B * rate))) if B >= 64 else 0
    rs = torch.full((B,), 1.0 / (1.0 - rate), device=device)
    if n_drop:
        rs[:: max(1, B // n_drop)][:n_drop] = 0.0
    kept = float((rs != 0).sum().item()) / B
    # the top instantiation of the step's kernel trace (gemm_tn_dma_kernel<128, 128, 3, 2>) on its largest shape: the stage-3 fc2 weight gradient
    # dW2[320, 1280] += dY[M3, 320]^T G[M3, 1280], M3 = B * 384 tokens (+ the bias gradient as a column sum of dY)
    M3 = B * 384
    dy3 = torch.randn(M3, 320, device=device).to(bf)
    g3 = torch.randn(M3, 1280, device=device).to(bf)
    dw3, db3 = torch.zeros(320, 1280, device=device), torch.zeros(320, device=device)
    # ... the way the step launches it (schedule.py: partials= + deferred fold): bf16 partial tiles into the scratch, ONE ordered fold behind it -- both inside the timed
    # closure; the GEMM's name is read between the two launches (round 5 timed the fp32-atomic path here, which the step no longer runs)
    scr3 = torch.empty(64 * 8 * 65536, dtype=bf, device=device)
    tn_names = {}

    def tn_s3():
        from mvlt_amd._lib import last_kernel
        ops.gemm_tn(dy3, g3, dw3, M3, 320, 1280, 320, 1280, 1280, colsum=db3, partials=scr3, defer_fold=True)
        if "gemm" not in tn_names:
            tn_names["gemm"] = last_kernel()
        ops.tn_fold_flush(scr3)

    tn_s3.names = tn_names
    return [("conv192", lambda: ops.gemm_nt(x, w, out, M, C, 9 * C, C, 9 * C, C, a_map=amap), 2.0 * M * C * 9 * C),
            ("proj64", lambda: ops.gemm_nt(x2, w2, o2, M2, 64, 64, 64, 64, 64, bias=b2), 2.0 * (2 * M2 * 64 + 64 * 64)),
            ("mlp_dw64", lambda: ops.mlp_bwd_dw(xm, dym, w1, w2t, b1, dw1, db1, dw2, db2, M2, 64, hid, row_scale=rs, rows_per_scale=4224),
             2.0 * 2 * M2 * 64 * hid * kept),
            ("tn_s3dw2", tn_s3, 2.0 * M3 * 320 * 1280)]


# what is KNOWN about a kernel instantiation (counters of earlier rounds): attached to a roofline entry only when the library reports that very
# instantiation for the timed call -- after a dispatch change the entry says so instead of repeating prose about a kernel that no longer runs
LIMITERS = {
    "mlp_wgrad2_kernel<64, 4>": "VALU + MFMA time add up: 9.5 VALU instructions + one LDS gather per (token, hidden unit) for GELU and GELU' (table over the "
                                "bf16 pre-activation) next to 4 x 64 MACs on the matrix pipe, at two waves per SIMD (216 registers)",
    "conv3_nt_kernel<32, 192, 1>": "LDS-DMA issue + MFMA: 128 x 192 tile, 3x3-gather A from an LDS halo (MFMA-busy 0.48)",
    "gemm_nt_dma_kernel<64, 0, 1, 64, 128>": "HBM: K = 64, 128 x 64 tile, every operand byte read once",
    "gemm_tn_p8_kernel<3, 3, 2, false, true, true>": "MFMA / LDS-DMA: 192 x 320 tiles of the 8-phase TN loop (seven row tiles for 6.67 of work, 224 of 256 CUs: whole m-splits per XCD), operands "
                                                     "swapped and the bf16 partial tiles stored transposed; 32 m-splits folded by one ordered pass (timed with it)",
    "gemm_tn_dma_kernel<128, 128, 3, 2, false>": "neither HBM- nor MFMA-bound (2.8 TB/s, MFMA-busy 0.34; 11-13 TB/s of L2 -> LDS-DMA requests): 128 x 128 tiles behind a 2-stage "
                                                 "LDS-DMA ring, two workgroups per CU, 16 m-splits leaving as bf16 partial tiles + one ordered fold (timed with it)",
}


def time_dominant_kernel(model, B, device, ms_step):
    """`roofline` = the launch VERDICT r3 named: the fused-MLP weight gradients of a stage-1 block (two per step), timed alone with HIP events on
    torch's current stream = the stream the C ABI launches on.  Its ALGORITHMIC work is the two weight-gradient products dW1 = dh^T x and
    dW2 = dy^T g: 2 x 2*M*C*hid FLOP (M = B*4224, C = 64, hid = 512) over 2 x M*C bf16 operand bytes; the kernel EXECUTES twice that (h = x W1^T and
    dg = dy W2 are recomputed on chip so that nothing of size M x hid touches HBM).  `bound` is "mfma" in the contract's vocabulary: the fraction says
    how far the launch is from doing its algorithmic FLOPs at matrix-pipe speed.  `siblings`: round 2's roofline launch (the MIM decoder's 192->192
    conv3x3, MFMA-bound), the HBM-bound K = 64 projection, and the top instantiation of the step's kernel trace (the 128 x 128 weight-gradient GEMM
    on the stage-3 fc2 shape).  Every `kernel` string is what the LIBRARY reports it launched for the timed call (mvlt_last_kernel), `share_of_step` is
    computed from the measured launch and step times, `limiter` is attached only if it was written for that instantiation.
    `traffic` = HBM bytes per launch from the committed PMC passes (profiles/*_roofline_traffic.json: FETCH_SIZE x2 + WRITE_SIZE, collected
    offline: counters cannot be read inside this process)."""
    cases = {name: (fn, work) for name, fn, work in roofline_cases(B, device)}
    timed = {name: _time_launch(fn) for name, (fn, _) in cases.items()}
    from mvlt_amd.build import source_hash
    tj = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.endswith("_roofline_traffic.json")) if B == 256 else []
    t = json.load(open(os.path.join(ROOT, "profiles", tj[-1]))) if tj else {}
    # the counters cannot be read inside this process: the committed pass is used only if it was collected for THESE kernel sources
    stale = None
    if t and t.get("_source_hash") != source_hash():
        stale = f"profiles/{tj[-1]} was collected for kernel sources {t.get('_source_hash')}, this tree is {source_hash()}: traffic not reported"
        t = {}
    tr = lambda k: t.get(k, {}).get("hbm_bytes")
    M2, M3 = B * 4224, B * 384

    def entry(name, what, per_step, bound, alg_bytes=None, executed=None):
        ms, kern = timed[name]
        names = getattr(cases[name][0], "names", None)
        if names and "gemm" in names:                  # two launches inside the timed closure: the GEMM's name + the fold's
            kern_all, kern = f"{names['gemm']} + {kern}", names["gemm"]
        else:
            kern_all = kern
        work = cases[name][1]
        e = dict(kernel=f"{kern_all} (bf16): {what}", kernel_reported_by="mvlt_last_kernel() after the timed launches",
                 share_of_step=f"{per_step} launch(es) per step x {ms:.4f} ms = {100 * per_step * ms / ms_step:.1f} % of the {ms_step:.2f} ms step",
                 bound=bound, ms_per_launch=round(ms, 4), traffic=tr(name))
        if bound == "mfma":
            tf = work / (ms * 1e-3) / 1e12
            e.update(achieved=round(tf, 1), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=round(tf / PEAK_BF16_TFLOPS, 4), algorithmic_flops=work)
            if executed:
                e["executed_flops"] = executed
            if alg_bytes:
                e["algorithmic_bytes"] = alg_bytes
        else:
            gbs = work / (ms * 1e-3) / 1e9
            e.update(achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4), algorithmic_bytes=work)
        e["limiter"] = LIMITERS.get(kern, f"not characterised: no counter analysis on record for {kern} (the dispatch chose another kernel than in earlier rounds)")
        return e

    top = entry("mlp_dw64", "fused-MLP weight gradients of a stage-1 block, M = B*4224 tokens, C = 64, hidden 512 (dW1, db1, dW2, db2; h / dg / GELU / GELU' "
                "recomputed on chip; the step's own DropPath factors: rate 0.1/7 of stage 1's second block, 4 of 256 samples dropped and skipped -- "
                "algorithmic_flops counts the kept samples only)", 2, "mfma", alg_bytes=2.0 * 2 * M2 * 64,
                executed=2 * cases["mlp_dw64"][1])
    # the same kernel INSIDE the step: average duration of that instantiation in the committed kernel trace of the bench command (rocprofv3 --kernel-trace --stats,
    # tools/profile_bench.sh -> profiles/rNN_kernel_stats.csv), whose launches are the two per step at DropPath rates 0 and 0.1/7 (+ the isolated timing's own)
    import csv
    import re
    ks = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if re.fullmatch(r"r\d+_kernel_stats\.csv", p)) if B == 256 else []
    if ks:
        kern_name = timed["mlp_dw64"][1]
        for row in csv.DictReader(open(os.path.join(ROOT, "profiles", ks[-1]))):
            if kern_name in row["Name"]:
                us = float(row["AverageNs"]) / 1e3
                full = 2.0 * 2 * M2 * 64 * 512                    # every sample kept (rate 0) -- the in-step launches drop at most 4 of 256
                top["frac_in_step"] = dict(frac=round(full / (us * 1e-6) / 1e12 / PEAK_BF16_TFLOPS, 4), avg_us=round(us, 1), calls=int(row["Calls"]),
                                           source=f"profiles/{ks[-1]} (a kernel trace of an earlier build when this file predates the tree: the per-round collection rewrites it)")
                break
    top["siblings"] = [
        entry("conv192", "MIM conv3x3 192->192 @32x32 as a gathered GEMM (M = B*1024, N = 192, K = 1728) -- round 2's roofline launch; 4 such forward / input-gradient "
              "launches per step", 4, "mfma"),
        entry("proj64", "K = 64, N = 64 projection with bias, M = B*4224 (the stage-1 q / proj forward)", 4, "hbm"),
        entry("tn_s3dw2", f"weight gradient dW2[320, 1280] += dY^T G over M = B*384 = {M3} rows + bias gradient: the stage-3 fc2 shape (round 5: the largest launch of the "
              "trace's top instantiation; stage-3 dW1 / dW2: 4 launches per step), partial tiles + fold as the step launches it", 4, "mfma",
              alg_bytes=2.0 * (M3 * 320 + M3 * 1280) + 4.0 * 320 * 1280)]
    top.update({"traffic_stale": stale} if stale else {"traffic_source": f"profiles/{tj[-1]}" if tj else None})
    return top


def step_traffic():
    """HBM GB per step of the committed FETCH_SIZE / WRITE_SIZE passes over whole steps (tools/step_traffic.sh -> profiles/rNN_step_traffic.txt);
    (None, why) when that pass was collected for other kernel sources than this tree's"""
    import re
    from mvlt_amd.build import source_hash
    fs = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.endswith("_step_traffic.txt"))
    if not fs:
        return None, None
    head = open(os.path.join(ROOT, "profiles", fs[-1])).readline()
    m = re.search(r"([0-9.]+) GB", head)
    h = re.search(r"source_hash=([0-9a-f]+)", head)
    if not h or h.group(1) != source_hash():
        return None, f"{fs[-1]} is stale (collected for kernel sources {h.group(1) if h else 'unknown'}, this tree is {source_hash()})"
    return (float(m.group(1)) if m else None), fs[-1]


def other_configs(device):
    """BASELINE configurations #4 and #5 at one GPU, behind the headline's timed region: short runs of the same engine entry
    -- PVT-medium at 384 px, batch 64 (5 + 10 iterations), and the CLS-head fine-tune step of pvlt_tiny at batch 256 (5 + 20; round 4 warmed it up for 20
    iterations on the belief of a 15 % ramp: timed per iteration it is flat from the fifth on, profiles/r05_ft_ramp.txt) -- so that the driver's own record carries them."""
    from mvlt_amd import pvlt
    from mvlt_amd.engine import BF16Scaler
    from mvlt_amd.optim import FusedAdamW
    import engine_grid_masking as E
    out = {}
    for key, name, img, B, lt, warm, steps, gflop in (
            ("medium384_b64", "pvlt_medium", 384, 64, dict(mlm=1, itm=1, t2i=1, cls=0), 5, 10, 197.25),
            ("finetune", "pvlt_tiny", 256, 256, dict(mlm=0, itm=0, t2i=0, cls=1), 5, 20, 25.00)):
        torch.manual_seed(4321)
        model = getattr(pvlt, name)(pretrained=False, token_hidden_size=768, num_text_tokens=128, loss_type=lt, pretrained_pth=None,
                                    drop_path_rate=0.1, drop_rate=0.0, num_classes=1000, in_chans=3).cuda(device)
        batch = synth_batch(B, img, 128, device, 99)
        opt = FusedAdamW(model, lr=2.5e-4 * B / 512.0, weight_decay=0.01)
        scaler = BF16Scaler()
        eargs = argparse.Namespace(loss_type=lt)

        def epoch(n, ep):
            with contextlib.redirect_stdout(sys.stderr):
                return E.train_one_epoch_vl(model, None, [batch] * n, opt, device, ep, scaler, None, None, None, True, False, eargs)

        epoch(warm, 0)
        torch.cuda.synchronize()
        t0 = time.time()
        st = epoch(steps, 1)
        torch.cuda.synchronize()
        dt = time.time() - t0
        out[key] = {"workload": f"{name} MVLT " + ("pre-train (MLM+MIM+ITM)" if lt["mlm"] else "fine-tune (CLS heads)") + f", {img}x{img} + 128 tokens, batch {B}, bf16",
                    "pairs_s": round(B * steps / dt, 1), "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps, "warmup": warm,
                    "step_tflops_reference_equivalent": round(B * steps / dt * gflop / 1e3, 1),
                    "mfma_frac_reference_equivalent": round(B * steps / dt * gflop / 1e3 / PEAK_BF16_TFLOPS, 4), "epoch_avg_loss": round(st["total_loss"], 4)}
        if key == "finetune":
            del model, opt
            continue
        # the eval callers' forward (evaluate_vl: model.eval(), no_grad, masked-row MLM head; BatchNorms of the MIM decoder folded into their convs) on the
        # pre-train model at batch 64 of configuration #4 -- and below on the headline model at batch 256
        del model, opt, batch
        torch.cuda.empty_cache()
    torch.manual_seed(4321)
    lt = dict(mlm=1, itm=1, t2i=1, cls=0)
    model = pvlt.pvlt_tiny(pretrained=False, token_hidden_size=768, num_text_tokens=128, loss_type=lt, pretrained_pth=None, drop_path_rate=0.1,
                           drop_rate=0.0, num_classes=1000, in_chans=3).cuda(device).eval()
    batch = synth_batch(256, 256, 128, device, 99)

    def fwd(n):
        with torch.no_grad():
            for _ in range(n):
                model(batch["image"], batch["input_ids"], mlm_labels=batch["mlm_labels"])

    fwd(3)
    torch.cuda.synchronize()
    t0 = time.time()
    fwd(10)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 10
    n_sel = int((batch["mlm_labels"] != -1).sum())
    fl = (FWD_ALL - FWD_MLM) + FWD_MLM * n_sel / (256 * 128)                  # executed forward FLOPs per pair (MLM head on the selected rows)
    out["eval_forward"] = {"workload": "pvlt_tiny MVLT eval forward (evaluate_vl's model call: eval mode, no_grad, masked-row MLM head), 256x256 + 128 tokens, batch 256, bf16",
                           "pairs_s": round(256 / dt, 1), "ms_per_batch": round(1e3 * dt, 3),
                           "executed_gflop_per_pair": round(fl / 1e9, 2), "tflops_executed": round(256 / dt * fl / 1e12, 1),
                           "mfma_frac_executed": round(256 / dt * fl / 1e12 / PEAK_BF16_TFLOPS, 4),
                           "tflops_reference_equivalent": round(256 / dt * FWD_ALL / 1e12, 1)}
    del model, batch
    torch.cuda.empty_cache()
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes of this one, which has not touched the GPU
    (no HIP call, no torch.cuda.is_available(); a process that has must never exec another program on this pool), the way the
    reference is started (scripts_dws/dws_mvlt_exp21.sh:13-16: torch.distributed.launch --nproc_per_node=8 main_vl.py), and leave
    with the launcher's exit code.  Rank 0's JSON line goes to this process's stdout unchanged."""
    import socket
    import subprocess
    have = torch.cuda.device_count()                      # counts devices without initialising the runtime on this image
    if have < n:
        print(f"bench.py: --gpus {n} but this node exposes {have} GPU(s)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    if n == 1:
        env["MVLT_DP_FORCE_COLLECTIVES"] = "1"            # world size 1 through the launcher = the RCCL code path of N > 1
    argv = [a for a in sys.argv[1:] if a != "--spawn"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="pairs per GPU")
    ap.add_argument("--img", type=int, default=256)
    ap.add_argument("--model", default="pvlt_tiny")
    ap.add_argument("--task", default="pretrain", choices=["pretrain", "finetune"],
                    help="pretrain = {mlm,itm,t2i} (BASELINE configs 2-4); finetune = {cls} only (config 5, dws_mvlt_ft_exp48)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the isolated roofline launches (whole-step counter passes: tools/step_traffic.sh)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of BASELINE configurations #4 / #5 behind the headline")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--spawn", action="store_true", help="start the ranks through torch.distributed.run even for --gpus 1 (RCCL path at world size 1)")
    args = ap.parse_args()

    if (args.gpus > 1 or args.spawn) and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # MVLT_DP_FORCE_COLLECTIVES=1 (test switch): build the process group and the data-parallel wrapper at world size 1 too, so that
    # a one-GPU box runs the RCCL calls of the N > 1 path (launched through torch.distributed.run --nproc-per-node 1)
    use_pg = world > 1 or bool(os.environ.get("MVLT_DP_FORCE_COLLECTIVES"))
    if use_pg:
        dist.init_process_group("nccl", device_id=device)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from mvlt_amd import pvlt
    from mvlt_amd.dist import DataParallel
    from mvlt_amd.engine import BF16Scaler
    from mvlt_amd.optim import FusedAdamW
    import engine_grid_masking as E                        # the drop-in module path reference main_vl.py:198 imports

    torch.manual_seed(1234 + rank)
    loss_type = dict(mlm=1, itm=1, t2i=1, cls=0) if args.task == "pretrain" else dict(mlm=0, itm=0, t2i=0, cls=1)
    model = getattr(pvlt, args.model)(pretrained=False, token_hidden_size=768, num_text_tokens=128, loss_type=loss_type,
                                      pretrained_pth=None, drop_path_rate=0.1, drop_rate=0.0, num_classes=1000, in_chans=3)
    model.cuda(device)
    core = model
    if use_pg:
        model = DataParallel(model)
    B = args.batch
    batch = synth_batch(B, args.img, 128, device, 1234 + rank)       # resident in HBM before the timed region starts
    n_sel = int((batch["mlm_labels"] != -1).sum())
    lr = 2.5e-4 * B * world / 512.0                       # reference main_vl.py:306
    opt = FusedAdamW(core, lr=lr, weight_decay=0.01)      # before any forward, like main_vl.py:308
    scaler = BF16Scaler()
    eargs = argparse.Namespace(loss_type=loss_type)

    def epoch(n_iter, ep):
        with contextlib.redirect_stdout(sys.stderr):      # the loop's progress lines must not mix with the one JSON line
            return E.train_one_epoch_vl(model, None, [batch] * n_iter, opt, device, ep, scaler, None, None, None, True, False, eargs)

    if args.warmup:
        epoch(args.warmup, 0)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    stats = epoch(args.steps, 1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.time() - t0
    per_rank = [dt]
    if use_pg:
        t = torch.zeros(world, device=device, dtype=torch.float64)
        t[rank] = dt
        dist.all_reduce(t, op=dist.ReduceOp.SUM)          # every rank's own time; the job's time is the slowest rank's
        per_rank = t.tolist()
        dt = max(per_rank)
    pairs_s = B * world * args.steps / dt

    # blocks-only GPU time (SRAttention + MLP Blocks incl. their LayerNorms): two more iterations with HIP events around the
    # Block kernels of every stage, forward and backward, outside the timed region
    core._block_events = []
    epoch(2, 2)
    torch.cuda.synchronize()
    ev, core._block_events = core._block_events, None
    blocks_ms = sum(a.elapsed_time(b) for a, b in zip(ev[0::2], ev[1::2])) / 2.0

    if rank == 0:
        ms_step = 1e3 * dt / args.steps
        line = {
            "metric": "image-text pairs/sec/node, PVT-tiny MVLT pre-train step" if (args.model, args.task) == ("pvlt_tiny", "pretrain")
            else f"image-text pairs/sec/node, {args.model} MVLT {args.task} step", "value": round(pairs_s, 2), "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "ranks_seen": dist.get_world_size() if use_pg else 1,
            "per_rank_pairs_s": [round(B * args.steps / t, 1) for t in per_rank],
            "config": {"workload": f"{args.model} MVLT " + ("pre-train (MLM+MIM+ITM)" if args.task == "pretrain" else "fine-tune (CLS heads)") + f", {args.img}x{args.img} RGB + 128 tokens, "
                                   f"batch {B}/GPU (global {B * world}), train_one_epoch_vl: fwd+loss+bwd+allreduce+AdamW",
                       "global_batch": B * world, "parallelism": f"dp{world}", "optimizer": "fused AdamW (fp32 master)",
                       "entry": "engine_grid_masking.train_one_epoch_vl", "epoch_avg_loss": round(stats["total_loss"], 4),
                       "parity_note": "bf16 `itm_logits` are gated on class probabilities (<= 2e-2) and on the error against max(|logits|, the dot product's own scale), not on relative logit error (the fixtures' logits cancel up to 30-fold); the reference's own bf16 autocast is 3.8e-2 off its fp32 self on this output; every other output is under the flat 2e-2 (tests/test_model_gpu.py)"},
        }
        # SURVEY.md 8d train-step FLOPs per pair of the other BASELINE configurations (reference-equivalent = 3 x forward)
        other = {("pvlt_medium", 384, "pretrain"): 197.25e9, ("pvlt_tiny", 256, "finetune"): 25.00e9}.get((args.model, args.img, args.task))
        if other:
            line["flops"] = {"reference_equivalent_gflop_per_pair": other / 1e9,
                             "step_tflops_reference_equivalent": round(pairs_s / world * other / 1e12, 1),
                             "mfma_frac_reference_equivalent": round(pairs_s / world * other / 1e12 / PEAK_BF16_TFLOPS, 4)}
        if args.model == "pvlt_tiny" and args.img == 256 and args.task == "pretrain":
            executed = 3 * (FWD_ALL - FWD_MLM) + 3 * FWD_MLM * n_sel / (B * 128) + 2 * FWD_FC1_RECOMPUTED
            per_gpu = pairs_s / world
            line["flops"] = {
                "reference_equivalent_gflop_per_pair": FLOP_PER_PAIR_TRAIN / 1e9,
                "executed_gflop_per_pair": round(executed / 1e9, 2),
                "executed_note": f"MLM head on the {n_sel} selected rows of {B * 128} only (reference-equivalent share 18.76 GFLOP/pair); "
                                 "+2.32 GFLOP/pair fc1 recomputed by the fused-MLP backward passes of stages 1-2",
                "step_tflops_reference_equivalent": round(per_gpu * FLOP_PER_PAIR_TRAIN / 1e12, 1),
                "step_tflops_executed": round(per_gpu * executed / 1e12, 1),
                "mfma_frac_reference_equivalent": round(per_gpu * FLOP_PER_PAIR_TRAIN / 1e12 / PEAK_BF16_TFLOPS, 4),
                "mfma_frac_executed": round(per_gpu * executed / 1e12 / PEAK_BF16_TFLOPS, 4),
                "blocks_only": {"what": "SRAttention + MLP Blocks (LN1, q, sr, kv, attention, proj, LN2, fc1, GELU, fc2; fwd + bwd): "
                                        "3 x 8.003 GFLOP/pair over the GPU time between HIP events around the Block kernels",
                                "ms_per_step": round(blocks_ms, 3),
                                "tflops": round(3 * FWD_BLOCKS * B / (blocks_ms * 1e-3) / 1e12, 1),
                                "mfma_frac": round(3 * FWD_BLOCKS * B / (blocks_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                "target": 0.40},
            }
            if not args.no_roofline:
                line["roofline"] = time_dominant_kernel(core, B, device, ms_step)
            gb, src = step_traffic()
            line["step"] = {"executed_tflops": round(per_gpu * executed / 1e12, 1), "ms": round(ms_step, 3),
                            "hbm_gb_per_step": gb, "hbm_tb_per_s": round(gb / ms_step, 2) if gb else None, "hbm_source": src}
            if world == 1 and B == 256 and not args.no_other_configs and not args.no_roofline:
                line["other_configs"] = other_configs(device)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if use_pg:
        dist.barrier()                                    # rank 0 is still timing its roofline launches: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
