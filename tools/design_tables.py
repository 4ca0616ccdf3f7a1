#!/usr/bin/env python3
"""Regenerates the measured tables of DESIGN.md section 3 (3.1: one step by kernel family, 3.2: the MFMA-carrying launches) from the committed
profiles of a round, so that the document cannot drift from the files it cites:
    python tools/design_tables.py r04            # rewrites the text between the <!-- BEGIN:3.x --> / <!-- END:3.x --> markers of DESIGN.md
    python tools/design_tables.py r04 --check    # exit 1 if the document and the profiles disagree"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
tag = args[0] if args else "r04"
CHECK = "--check" in sys.argv          # exit 1 if DESIGN.md differs from what the profiles say (tests/test_host_cpu.py)
P = lambda name: os.path.join(ROOT, "profiles", f"{tag}_{name}")

# ---------------------------------------------------------------- 3.1
head = open(P("step_launches.txt")).readline()
m = re.search(r"(\d+) launches, ([\d.]+) ms .*kernels busy ([\d.]+) ms \(([\d.]+) %\); ATen kernels (\d+), __amd_rocclr_copyBuffer (\d+)", head)
n_launch, span_ms, busy_ms, busy_pct, n_aten, n_copy = int(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(4)), int(m.group(5)), int(m.group(6))
fam = {}
for line in open(P("step_launches.txt")).read().split("\n")[2:]:
    mm = re.match(r"\s*(\d+)\s+([\d.]+)\s+(.*)$", line)
    if not mm:
        continue
    n, us, name = int(mm.group(1)), float(mm.group(2)), mm.group(3)
    if "ln_bwd" in name or "ln_fwd" in name:
        k = "ln"
    elif any(x in name for x in ("bn_", "col_reduce", "upsample", "ew_mul")):
        k = "mim"
    elif "at::" in name or "rocclr" in name:
        k = "aten"
    else:
        if name.startswith(("gemm_tn_p8", "tn_fold")):
            name = "gemm_tn_dma(p8)" + name
        k = next((x for x in ("gemm_tn_dma", "gemm_nt_p8", "gemm_nt_dma", "mlp_pipe", "mlp_wgrad2", "attn_bwd", "attn_fwd2", "conv3_nt", "conv3_wgrad", "adamw",
                              "bert_embed_bwd", "bert_embed_fwd", "weight_prep", "ce_fwd", "ce_bwd", "fold_copies") if name.startswith(x)), "other")
    a = fam.setdefault(k, [0, 0.0])
    a[0] += n
    a[1] += us
total = sum(v[1] for v in fam.values())
_rf = json.loads(open(P("bench_n1.json")).read().strip().split("\n")[-1])["roofline"]["frac"]
row = lambda k: (fam.get(k, [0, 0.0])[0], fam.get(k, [0, 0.0])[1], 100.0 * fam.get(k, [0, 0.0])[1] / total)
R = {}
for k in fam:
    R[k] = row(k)
rest_keys = [k for k in fam if k in ("adamw", "bert_embed_bwd", "bert_embed_fwd", "weight_prep", "ce_fwd", "ce_bwd", "fold_copies", "other", "aten")]
rest_n, rest_us = sum(fam[k][0] for k in rest_keys), sum(fam[k][1] for k in rest_keys)
g = lambda k: f"{R[k][0]} | {R[k][1]:.0f} | {R[k][2]:.1f} %"
us = lambda k: fam.get(k, [0, 0.0])[1]
prev_tag_early = "r%02d" % (int(tag[1:]) - 1)
t31 = f"""### 3.1 One steady-state step: {n_launch} launches, kernels busy {busy_ms:.2f} ms (`profiles/{tag}_step_launches.txt`)

(The span from the step's first kernel to the optimizer's end is {span_ms:.2f} ms under `rocprofv3 --kernel-trace` -- {busy_pct:.1f} % busy; {n_aten} of the launches are ATen kernels, {n_copy} `copyBuffer`.
Unprofiled the bench step equals the sum of the kernel durations.  The boxes of the pool differ by up to 5 % (19.4-20.5 ms for this build): family differences to `{prev_tag_early}_step_launches.txt` below that are spread; the kernel changes of round 6 inside this step are in the weight-gradient family (the stage-3 fc gradients on 192 x 320 tiles of the 8-phase TN loop: -0.04 ms; the vocabulary decoder's gradient stored instead of added: -0.1 ms).)

| kernel family | launches | us / step | share | bound by (evidence) |
|---|---|---|---|---|
| `gemm_tn_dma_kernel` + `gemm_tn_p8_kernel` / `tn_fold_kernel` (weight gradients; 8 launches carry the input gradient too; the stage-4 MLP ones and every launch with >= 8 m-splits reduce through bf16 partial tiles + the batched fold (`tn_fold_multi_kernel`, 5 launches per step, the conv3x3 outputs included: counted here)) | {g('gemm_tn_dma')} | MFMA-bound shapes 0.24-0.41 of peak (3.2), the K <= 128 shapes HBM (4.5-5.6 TB/s); the 128 x 128 instantiation sits at 2.0 x its HBM time and 3.4 x its MFMA time behind 11-13 TB/s of L2 -> LDS-DMA requests (3.5); where atomics remain a split reduction costs outputs x splits / 0.3 ns (`experiments_r4.md` 2) |
| `gemm_nt_dma_kernel` (128-wide NT GEMMs: K <= 128 projections, gathers, small heads) | {g('gemm_nt_dma')} | HBM for K = 64 / 128 (`proj64` sibling: 0.60-0.70 of 8 TB/s); TA / L1 path for the rest (`r03_l1_stalls.txt`) |
| `gemm_nt_p8_kernel` (8-wave / 8-phase NT GEMMs, stage 3-4, MLM logits) | {g('gemm_nt_p8')} | K-loops alone 1.1-1.6 PFLOP/s (epilogue compiled out, `r05_p8_epilogue_ablation.txt`); the launches 0.23-0.57 of peak: whole-round quantisation + a VALU-bound epilogue (~15 instructions per output at two waves per SIMD) that a persistent grid does not hide (3.3, `experiments_r5.md` 3) |
| `mlp_pipe_kernel` (fused MLP forward / input gradient, stages 1-2) | {g('mlp_pipe')} | VALU (GELU: 10 instructions per hidden element) + MFMA, partly overlapped: MFMA-busy 0.28-0.35, VALU-active 0.26-0.35 (`{tag}_mfma_counters.csv`) |
| `mlp_wgrad2_kernel` (fused MLP weight gradients) | {g('mlp_wgrad2')} | VALU + MFMA issue times ADD on a SIMD (1.3 x their sum; `roofline`: {_rf:.3f} algorithmic / {2 * _rf:.2f} executed); neither exposed LDS-gather latency nor two-against-four waves per SIMD moves it (`experiments_r6.md` 3) |
| LayerNorm forward / backward (standalone launches) | {g('ln')} | HBM + Infinity Cache: 4.5-7.8 TB/s algorithmic (streaming passes over the fp32 residual stream); round 3 / first half of round 4: 2238 us (`experiments_r4.md` 7) |
| `conv3_nt_kernel` (MIM conv3x3 forward / dgrad) | {g('conv3_nt')} | MFMA / LDS-DMA: 1.0-1.24 PFLOP/s (0.41-0.50), MFMA-busy 0.48 |
| MIM decoder non-GEMM (BatchNorm, upsample, products, fused loss) | {g('mim')} | HBM streaming, fp16 z and product factors (first half of round 4: 1482 us, `experiments_r4.md` 8) |
| `attn_bwd_dma_kernel` | {g('attn_bwd')} | dependent chain per 32-query tile at two waves per SIMD (256 registers): MFMA-busy 0.23, waits 0.37 + 0.26; stage 1 = 2.4 x its HBM floor; a split into dQ + dK / dV kernels cannot win (`experiments_r5.md` 2); prologue + flush 21 % / 9 % of the launch at stages 4 / 3 |
| `conv3_wgrad_kernel` (bf16 partial blocks, folded by `tn_fold_kernel`: round 4 with the atomic flush 1005 us) | {g('conv3_wgrad')} | MFMA / LDS: the loop alone ran at ~1.5 PFLOP/s in the flush ablation (`r05_conv_wgrad_atomics_ablation.txt`) |
| `attn_fwd2_kernel` | {g('attn_fwd2')} | Q / O streaming at stage 1 (77-80 us vs ~55 us floor), K / V staging at stages 3-4 |
| AdamW {us('adamw'):.0f}, BERT-embedding bwd / fwd {us('bert_embed_bwd'):.0f} / {us('bert_embed_fwd'):.0f}, weight prep {us('weight_prep'):.0f}, cross entropy {us('ce_fwd') + us('ce_bwd'):.0f}, gradient-copy folds {us('fold_copies'):.0f}, ATen leftovers ({R.get('aten', (0, 0, 0))[0]} launches) {us('aten'):.0f}, other helpers | {rest_n} | {rest_us:.0f} | {100 * rest_us / total:.1f} % | HBM (AdamW: 1.2 GB at 5.7-6.4 TB/s; the embedding backward: 25 M fp32 atomics at 0.3 per ns) |
"""

# ---------------------------------------------------------------- 3.2
shapes = {}
for line in open(P("gemm_shapes.txt")):
    mm = re.match(r"\s*([\d.]+) ms\s+x\s+(\d+)\s+([\d.]+) us\s+(?:([\d.]+) TF/s\s+)?(.*)$", line)
    if mm:
        shapes[" ".join(mm.group(5).split())] = (int(mm.group(2)), float(mm.group(3)), float(mm.group(4)) if mm.group(4) else None)
tot_line = [l for l in open(P("gemm_shapes.txt")) if l.startswith("total")][0].strip()


def S(key):
    for k, v in shapes.items():
        if k.startswith(key):
            return v
    raise KeyError(key)


# the previous round's time of the same launch, from ITS committed per-shape file (last column of 3.2)
prev_tag = "r%02d" % (int(tag[1:]) - 1)
prev_shapes = {}
_pp = os.path.join(ROOT, "profiles", f"{prev_tag}_gemm_shapes.txt")
if os.path.exists(_pp):
    for _l in open(_pp):
        mm = re.match(r"\s*([\d.]+) ms\s+x\s+(\d+)\s+([\d.]+) us\s+(?:([\d.]+) TF/s\s+)?(.*)$", _l)
        if mm:
            prev_shapes[" ".join(mm.group(5).split())] = float(mm.group(3))


def PV(*keys):
    out = []
    for key in keys:
        v = next((u for k, u in prev_shapes.items() if k.startswith(key)), None)
        out.append("-" if v is None else f"{v:.1f}")
    return " / ".join(out)


def r(key, flops=None):
    n, us_, tf = S(key)
    if tf is None and flops:
        tf = flops / us_ / 1e6
    return us_, tf


def line(label, kern, key, floor, r3, flops=None):
    us_, tf = r(key, flops)
    return f"| {label} | {kern} | {us_:.1f} | {tf:.0f} | {tf / 2500:.2f} | {floor} | {PV(key)} |"


def line2(label, kern, key_a, key_b, floor, r3):
    (ua, ta), (ub, tb) = r(key_a), r(key_b)
    return f"| {label} | {kern} | {ua:.1f} / {ub:.1f} | {ta:.0f} / {tb:.0f} | {ta / 2500:.2f} / {tb / 2500:.2f} | {floor} | {PV(key_a, key_b)} |"


rows = [
    line("stage-3 fc1 + GELU 98304 x 1280 x 320 (H and G stored)", "p8 256 x 256", "gemm_nt 98304 1280 320 A:- C:- b act1", 91, 137.5),
    line2("stage-3 fc2 + fp32 residual 98304 x 320 x 1280 (first block: fp32 out / last block: bf16 out, `r_fp32`)", "p8 192 x 320", "gemm_nt 98304 320 1280 A:- C:- b act0 R float32",
          "gemm_nt 98304 320 1280 A:- C:- b act0 R bfloat16", "80 / 70", 155.1),
    line("stage-3 GELU' dgrad 98304 x 1280 x 320", "p8 256 x 256", "gemm_nt 98304 1280 320 A:- C:- act2", 90, 176.0),
    line("stage-3 fc1 dgrad 98304 x 320 x 1280", "p8 192 x 320", "gemm_nt 98304 320 1280 A:- C:- act0 bfloat16", 50, 106.2),
]
a, b = r("gemm_tn 98304 1280 320"), r("gemm_tn 98304 320 1280")
rows.append(f"| stage-3 dW1 / dW2 (TN, 98304 rows; round 6: 192 x 320 tiles of the 8-phase loop, ragged last row tile, bf16 partial tiles + fold) | tn p8 192 x 320 + fold | {a[0]:.1f} / {b[0]:.1f} | {a[1]:.0f} / {b[1]:.0f} | {a[1] / 2500:.2f} / {b[1] / 2500:.2f} | 50 | {PV('gemm_tn 98304 1280 320', 'gemm_tn 98304 320 1280')} |")
rows += [
    line("stage-4 fc1 + GELU 49152 x 2048 x 512", "p8 256 x 256", "gemm_nt 49152 2048 512 A:- C:- b act1", 72, 150.3),
    line2("stage-4 fc2 + fp32 residual 49152 x 512 x 2048 (fp32 out / bf16 out)", "p8 192 x 256", "gemm_nt 49152 512 2048 A:- C:- b act0 R float32",
          "gemm_nt 49152 512 2048 A:- C:- b act0 R bfloat16", "64 / 56", 139.6),
    line("stage-4 GELU' dgrad 49152 x 2048 x 512", "p8 256 x 256", "gemm_nt 49152 2048 512 A:- C:- act2", 72, 168.7),
    line("stage-4 fc1 dgrad 49152 x 512 x 2048", "p8 192 x 256", "gemm_nt 49152 512 2048 A:- C:- act0 bfloat16", 40, 91.8),
]
a, b = r("gemm_tn 49152 2048 512"), r("gemm_tn 49152 512 2048")
rows.append(f"| stage-4 dW2 / dW1 (TN, 49152 rows; bf16 partial tiles + fold, no atomics) | tn p8 256 x 256 + fold | {a[0]:.1f} / {b[0]:.1f} | {a[1]:.0f} / {b[1]:.0f} | {a[1] / 2500:.2f} / {b[1] / 2500:.2f} | 40 | {PV('gemm_tn 49152 2048 512', 'gemm_tn 49152 512 2048')} |")
TWELVE = ("gemm_nt 98304 1280 320 A:- C:- b act1", "gemm_nt 98304 320 1280 A:- C:- b act0 R float32", "gemm_nt 98304 1280 320 A:- C:- act2",
                               "gemm_nt 98304 320 1280 A:- C:- act0 bfloat16", "gemm_tn 98304 1280 320", "gemm_tn 98304 320 1280",
                               "gemm_nt 49152 2048 512 A:- C:- b act1", "gemm_nt 49152 512 2048 A:- C:- b act0 R float32", "gemm_nt 49152 2048 512 A:- C:- act2",
                               "gemm_nt 49152 512 2048 A:- C:- act0 bfloat16", "gemm_tn 49152 2048 512", "gemm_tn 49152 512 2048")
twelve = sum(r(k)[0] for k in TWELVE)
_pt = [next((u for kk, u in prev_shapes.items() if kk.startswith(k)), None) for k in TWELVE]
twelve_prev = f"{sum(_pt):.0f}" if all(v is not None for v in _pt) else "-"
rows.append(f"| **the twelve MLP launches of a stage-3 + a stage-4 block** | | **{twelve:.0f}** | | | | **{twelve_prev}** (VERDICT r4 target 1200: not reachable with an f32 GELU epilogue on this tile, 3.3) |")
cf, cd, cw = r("gemm_nt 262144 192 1728 A:m2r3 C:- act0 float16"), r("gemm_nt 262144 192 1728 A:m2r3 C:- act0 bfloat16"), r("gemm_tn 262144 192 1728")
rows.append(f"| MIM conv3x3 192 -> 192 @ 32 x 32 forward (fp16 z + BN statistics) / dgrad (bf16) | conv3_nt | {cf[0]:.1f} / {cd[0]:.1f} | {cf[1]:.0f} / {cd[1]:.0f} | {cf[1] / 2500:.2f} / {cd[1] / 2500:.2f} | 40 | {PV('gemm_nt 262144 192 1728 A:m2r3 C:- act0 float16', 'gemm_nt 262144 192 1728 A:m2r3 C:- act0 bfloat16')} |")
rows.append(f"| its weight gradient | conv3_wgrad | {cw[0]:.1f} | {cw[1]:.0f} | {cw[1] / 2500:.2f} | 32 | {PV('gemm_tn 262144 192 1728')} |")
rows.append(line("MLM logits 1490 x 30522 x 768 (fp32 out)", "p8 256 x 256, ragged", "gemm_nt 1490 30522 768", 37, 157.6))
GF1, GF2 = 2.0 * 1081344 * 64 * 512, 2.0 * 294912 * 128 * 1024       # one GEMM unit of the fused MLP
f1, x1, w1 = S("mlp_fwd 1081344 64 512")[1], S("mlp_bwd_dx 1081344 64 512")[1], S("mlp_bwd_dw 1081344 64 512")[1]
f2, x2, w2 = S("mlp_fwd 294912 128 1024")[1], S("mlp_bwd_dx 294912 128 1024")[1], S("mlp_bwd_dw 294912 128 1024")[1]
tfs = lambda units, gf, us_: units * gf / us_ / 1e6
rows.append(f"| fused MLP stage 1 (M = 1081344, C = 64, hidden 512): forward / input gradient (+ `norm2` backward) / weight gradients | mlp_pipe / mlp_wgrad2 | {f1:.1f} / {x1:.1f} / {w1:.1f} | "
            f"{tfs(2, GF1, f1):.0f} / {tfs(3, GF1, x1):.0f} / {tfs(4, GF1, w1):.0f} executed | {tfs(2, GF1, f1) / 2500:.2f} / {tfs(3, GF1, x1) / 2500:.2f} / {tfs(4, GF1, w1) / 2500:.2f} executed | 50-100 | {PV('mlp_fwd 1081344 64 512', 'mlp_bwd_dx 1081344 64 512', 'mlp_bwd_dw 1081344 64 512')} |")
rows.append(f"| fused MLP stage 2 (M = 294912, C = 128, hidden 1024) | same | {f2:.1f} / {x2:.1f} / {w2:.1f} | {tfs(2, GF2, f2):.0f} / {tfs(3, GF2, x2):.0f} / {tfs(4, GF2, w2):.0f} executed | "
            f"{tfs(2, GF2, f2) / 2500:.2f} / {tfs(3, GF2, x2) / 2500:.2f} / {tfs(4, GF2, w2) / 2500:.2f} executed | 30-60 | {PV('mlp_fwd 294912 128 1024', 'mlp_bwd_dx 294912 128 1024', 'mlp_bwd_dw 294912 128 1024')} |")
ab = [S(f"sr_attention_bwd 256 {h} {n} 192")[1] for h, n in ((1, 4224), (2, 1152), (5, 384), (8, 192))]
af = [S(f"sr_attention_fwd 256 {h} {n} 192")[1] for h, n in ((1, 4224), (2, 1152), (5, 384), (8, 192))]
fl = [256 * h * n * 192 * 64 * 2.0 for h, n in ((1, 4224), (2, 1152), (5, 384), (8, 192))]
rows.append("| SR attention backward, stages 1 / 2 / 3 / 4 | attn_bwd_dma | " + " / ".join(f"{x:.1f}" for x in ab) + " | " + " / ".join(f"{5 * f / x / 1e6:.0f}" for f, x in zip(fl, ab)) + " | " +
            " / ".join(f"{5 * f / x / 1e6 / 2500:.2f}" for f, x in zip(fl, ab)) + " | 95 / 56 / 60 / 64 | " + PV(*[f"sr_attention_bwd 256 {h} {n} 192" for h, n in ((1, 4224), (2, 1152), (5, 384), (8, 192))]) + " |")
rows.append("| SR attention forward, stages 1-4 | attn_fwd2 | " + " / ".join(f"{x:.1f}" for x in af) + " | " + " / ".join(f"{2 * f / x / 1e6:.0f}" for f, x in zip(fl, af)) + " | " +
            " / ".join(f"{2 * f / x / 1e6 / 2500:.2f}" for f, x in zip(fl, af)) + " | 46 / 28 / 30 / 32 | " + PV(*[f"sr_attention_fwd 256 {h} {n} 192" for h, n in ((1, 4224), (2, 1152), (5, 384), (8, 192))]) + " |")
d1, d2 = S("gemm_tn 1081344 64 64")[1], S("gemm_tn 294912 128 128")[1]
b1, b2 = (1081344 * 64 * 2 * 3 + 0.0) / d1 / 1e6, (294912 * 128 * 2 * 3 + 0.0) / d2 / 1e6
rows.append(f"| q / proj weight + input gradient in one pass, stage 1 (1081344 x 64 x 64) / stage 2 (294912 x 128 x 128) | tn 64 x 64 / 128 x 128 + DG | {d1:.1f} / {d2:.1f} | HBM: {b1:.1f} / {b2:.1f} TB/s | "
            f"{b1 / 8:.2f} / {b2 / 8:.2f} of 8 TB/s | 66 / 36 | {PV('gemm_tn 1081344 64 64', 'gemm_tn 294912 128 128')} |")
bench = json.loads(open(P("bench_n1.json")).read().strip().split("\n")[-1])
bo = bench["flops"]["blocks_only"]
t32 = f"""### 3.2 The MFMA-carrying launches (`profiles/{tag}_gemm_shapes.txt`: every distinct launch of the step re-timed alone with the step's arguments; {tot_line})

fraction = 2 M N K / time / 2.5 PFLOP/s; "HBM floor" = algorithmic bytes / 6.3 TB/s (attention: Q, dO, O, dQ + K / V + dK / dV + lse in bf16 -- round 4's table
had left the K / V / dK / dV streams out of the stage 2-4 floors and showed 27 / 15 us there).

| launch (M x N x K, epilogue) | kernel / tile | us | TFLOP/s | fraction | HBM floor us | round {int(tag[1:]) - 1} us (`profiles/{prev_tag}_gemm_shapes.txt`, another box) |
|---|---|---|---|---|---|---|
""" + "\n".join(rows) + f"""

Blocks-only MFMA fraction (north_star's figure, `flops.blocks_only` of the bench line): **{bo['mfma_frac']:.3f}** ({bo['ms_per_step']:.2f} ms for 6.15 TFLOP; round 5: 0.182 driver, round 4: 0.180, round 3: 0.165; target 0.40
-- 3.5 totals what this kernel decomposition could reach if every launch ran at the first limit it meets: ~0.30).
"""

# ---------------------------------------------------------------- 6: the line of the final build
rf, fl_, st, cb, oc = bench["roofline"], bench["flops"], bench["step"], bench["cpu_baseline"], bench.get("other_configs", {})
sib = rf["siblings"]
tj = json.load(open(P("roofline_traffic.json")))
ks = [l for l in open(P("kernel_stats.csv")) if "mlp_wgrad2_kernel<64" in l][0].rsplit('",', 1)[1].split(",")
ks_calls, ks_avg_us = int(ks[0]), float(ks[2]) / 1e3
mw, cal = tj["mlp_dw64"], tj["calib_cast"]
pairs = lambda x: f"{x:,.0f}".replace(",", " ")
_fis = (rf.get("frac_in_step") or {}).get("frac", float("nan"))
t6 = f"""**The line of the final build** (`profiles/{tag}_bench_n1.json`, command `python bench.py`, sources `{tj['_source_hash']}`): **{pairs(bench['value'])} pairs/s, {bench['ms_per_step']:.2f} ms/step**
(round 5: driver 12 872 / 19.89; round 4: 12 532 / 20.43; round 3: 11 428 / 22.40).  The boxes of the pool differ by +-2.5 %: this tree measured 19.4 .. 20.2 ms on the boxes of this
round.  Round 6 changed two hot paths of this step (the stage-3 fc weight gradients: -0.04 ms same-box; the vocabulary decoder's weight gradient without atomics: -0.1 ms; `docs/experiments_r6.md`: seven other kernel-level attempts measured same-box, none faster), so the step is round 5's within that spread; what changed in the LINE:
the roofline launch runs with the step's own DropPath factors (4 of 256 samples dropped, `algorithmic_flops` counts the kept ones -- round 5 dropped one in ten while counting all: the `frac` erred upward), `frac_in_step` gives the same kernel's fraction from the
committed kernel trace, the TN sibling is timed the way the step launches it (partial tiles + fold), `other_configs` are faster (pvlt_medium at 384 px: ragged 192 x 320 tiles + one-chunk attention backward), `config.parity_note` states the bf16 ITM exemption.
Loss trajectory unchanged (epoch average {bench['config']['epoch_avg_loss']:.2f}, same synthetic batch).

| field | value | how to recompute it |
|---|---|---|
| `value`, `ms_per_step` | {pairs(bench['value'])} pairs/s, {bench['ms_per_step']:.3f} ms | 256 pairs x 20 steps / wall time between `torch.cuda.synchronize()`s around `train_one_epoch_vl`; kernel sum of one step: {busy_ms:.2f} ms (`{tag}_step_launches.txt`) |
| `flops.blocks_only` | {bo['ms_per_step']:.2f} ms, {bo['tflops']:.1f} TFLOP/s, **{bo['mfma_frac']:.3f}** | HIP events around the Block kernels of every stage, fwd + bwd, two extra iterations; 3 x 8.003 GFLOP x 256 / {bo['ms_per_step']:.2f} ms / 2.5 PFLOP/s |
| `flops.mfma_frac_executed` | {fl_['mfma_frac_executed']:.3f} | {fl_['executed_gflop_per_pair']:.2f} GFLOP/pair executed (MLM head on the selected rows of 32768 only; + 2.32 fc1 recomputed) x {pairs(bench['value'])} / 2.5 PFLOP/s |
| `roofline` (kernel as reported by the library: `{rf['kernel'].split(' (bf16)')[0]}`, the launch VERDICT r3 named) | achieved {rf['achieved']:.1f} TFLOP/s, **frac {rf['frac']:.3f}**, {rf['ms_per_launch']:.3f} ms; {rf['share_of_step']} | algorithmic FLOPs 2 x 2 M C hid x kept samples = {rf['algorithmic_flops'] / 1e9:.2f} GFLOP / {rf['ms_per_launch']:.3f} ms (HIP events, 20 launches, torch's current stream = the launch stream); `{tag}_kernel_stats.csv`: {ks_avg_us:.1f} us average over {ks_calls} calls (in-step + this timing) -> {141.73e9 / (ks_avg_us * 1e-6) / 1e12:.1f} TFLOP/s = {141.73e9 / (ks_avg_us * 1e-6) / 1e12 / 2500:.3f}; executed FLOPs are twice the algorithmic ones (h and dg recomputed on chip); `frac_in_step` of the line: {_fis:.3f} |
| `roofline.traffic` | {rf['traffic'] / 1e6:.1f} MB per launch | `{tag}_roofline_traffic.json`: 2 x FETCH_SIZE ({mw['fetch_size_kib']:,.0f} KiB) + WRITE_SIZE ({mw['write_size_kib']:,.0f} KiB), separate `--pmc` passes over `tools/roofline_launch.py`, {mw['dispatches']} dispatches; calibration in the same run: torch's fp32 -> bf16 cast of a 262144 x 192 tensor reads {cal['read_bytes'] / 1e6:.2f} MB (expected {cal['expected_read_bytes'] / 1e6:.2f}) and writes {cal['write_bytes'] / 1e6:.2f} ({cal['expected_write_bytes'] / 1e6:.2f}); algorithmic bytes 276.8 MB -> {rf['traffic'] / 276.824064e6:.2f} x (the partial-sum flushes) |
| `roofline.siblings` | conv3x3 192 -> 192 (`{sib[0]['kernel'].split(' (bf16)')[0]}`): {sib[0]['achieved']:.0f} TFLOP/s = **{sib[0]['frac']:.2f}**, {sib[0]['traffic'] / 1e6:.1f} MB; K = 64 projection (`{sib[1]['kernel'].split(' (bf16)')[0]}`): {sib[1]['achieved'] / 1e3:.2f} TB/s = **{sib[1]['frac']:.2f}** of 8 TB/s, {sib[1]['traffic'] / 1e6:.0f} MB; stage-3 fc2 weight gradient (`{sib[2]['kernel'].split(' (bf16)')[0]}`, the trace's top instantiation): {sib[2]['achieved']:.0f} TFLOP/s = **{sib[2]['frac']:.2f}**, {sib[2]['ms_per_launch'] * 1e3:.1f} us, {(sib[2]['traffic'] or 0) / 1e6:.1f} MB against {sib[2]['algorithmic_bytes'] / 1e6:.1f} MB algorithmic | {sib[0]['algorithmic_flops'] / 1e9:.1f} GFLOP / {sib[0]['ms_per_launch']:.4f} ms; {sib[1]['algorithmic_bytes'] / 1e6:.1f} MB / {sib[1]['ms_per_launch']:.4f} ms; {sib[2]['algorithmic_flops'] / 1e9:.1f} GFLOP / {sib[2]['ms_per_launch']:.4f} ms |
| `step.hbm_gb_per_step` | {st['hbm_gb_per_step']:.2f} GB, {st['hbm_tb_per_s']:.2f} TB/s | `{tag}_step_traffic.txt`: FETCH_SIZE x 2 + WRITE_SIZE over 6 whole steps of the bench process, per kernel (round 5: 63.21, round 4: 63.19, round 3: 66.65) |
| `cpu_baseline` | {cb['value']:.1f} pairs/s train step, {cb.get('forward_loss_value', 0):.1f} forward + loss, {cb['cores']} cores, `kind: {cb['kind']}` | the oracle at config #1 shapes (4 pairs, fp32), ~20 s sample on the box's host cores; a reported baseline, not a target |
| `other_configs` | medium384_b64 **{pairs(oc['medium384_b64']['pairs_s'])} pairs/s** ({oc['medium384_b64']['ms_per_step']:.1f} ms, {oc['medium384_b64']['mfma_frac_reference_equivalent']:.3f} of peak reference-equivalent), finetune **{pairs(oc['finetune']['pairs_s'])} pairs/s** ({oc['finetune']['ms_per_step']:.1f} ms, {oc['finetune']['mfma_frac_reference_equivalent']:.3f}), eval forward **{pairs(oc['eval_forward']['pairs_s'])} pairs/s** ({oc['eval_forward']['ms_per_batch']:.2f} ms per batch of 256, {oc['eval_forward']['mfma_frac_executed']:.3f} executed) | BASELINE configurations #4 / #5 at one GPU, 5 + 10 / 5 + 20 iterations of the same engine entry behind the headline; the eval callers' model call (eval mode, no_grad, masked-row MLM head, BatchNorms folded) 3 + 10 times |
"""

# ---------------------------------------------------------------- 3.5: what the step could take (tools/ceiling_table.py over the whole-step counter passes)
import subprocess
_ct = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ceiling_table.py"), P("step_sq.csv"), P("step_traffic.txt"), "--md", "--check"], capture_output=True, text=True)
if _ct.returncode != 0:
    print("tools/ceiling_table.py failed:", _ct.stdout[-400:], _ct.stderr[-400:])
    sys.exit(1)
t35 = f"""### 3.5 The ceiling of this kernel decomposition, totalled (`python tools/ceiling_table.py profiles/{tag}_step_sq.csv profiles/{tag}_step_traffic.txt --md`)

Per kernel of one step, from two whole-step counter passes over the bench process (`tools/step_sq.sh`, `tools/step_traffic.sh`; times in us per step): t_bytes = its measured HBM bytes / 6.3 TB/s,
t_mfma = its executed MFMA FLOPs (`SQ_INSTS_VALU_MFMA_MOPS_BF16` x 512) / 2.5 PFLOP/s, t_valu = its non-MFMA VALU wave-instructions / 1024 SIMDs x 1.5 ns (the measured issue rate of plain
f32 VALU at four waves per SIMD); ceiling = the largest of the three, i.e. perfect overlap of the three pipes inside every kernel and nothing else in the way.

""" + _ct.stdout + """
Read with `docs/experiments_r6.md` 3 and 5: on ONE SIMD the matrix and the vector pipe mostly serialise (a {MFMA + 8 VALU} stream costs 15.9 ns where the two alone cost 8.05 + 12.4; at two
waves per SIMD 28.7), so the fused-MLP kernels' real floor is the SUM of t_mfma and t_valu, not their maximum -- they run at 1.2-1.4 x that sum -- and the table's "ceiling" is an upper bound on what tuning can return.
"""

p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
for name, text in (("3.1", t31), ("3.2", t32), ("3.5", t35), ("6", t6)):
    b, e = f"<!-- BEGIN:{name} -->", f"<!-- END:{name} -->"
    i, j = s.index(b) + len(b), s.index(e)
    s = s[:i] + "\n" + text + s[j:]
if CHECK:
    if s != open(p).read():
        print("DESIGN.md sections 3.1 / 3.2 / 3.5 / 6 differ from profiles/%s_*: run python tools/design_tables.py %s" % (tag, tag))
        sys.exit(1)
    print("DESIGN.md sections 3.1 / 3.2 / 6 match profiles/%s_*" % tag)
    sys.exit(0)
open(p, "w").write(s)
print(f"DESIGN.md 3.1 / 3.2 rewritten from profiles/{tag}_*: {n_launch} launches, busy {busy_ms:.2f} ms, twelve MLP launches {twelve:.0f} us, blocks-only {bo['mfma_frac']:.3f}")
