#!/usr/bin/env python3
"""What the step could take if every kernel ran at the first limit it meets: per kernel of one steady-state training step

    t_bytes = HBM bytes of the kernel (FETCH_SIZE x 2 + WRITE_SIZE, tools/step_traffic.sh; >= its algorithmic bytes) / 6.3 TB/s (the rate streaming kernels reach here)
    t_mfma  = executed MFMA FLOPs (SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512) / 2.5 PFLOP/s
    t_valu  = non-MFMA VALU wave-instructions (SQ_INSTS_VALU - MOPS / 32) / 1024 SIMDs x 1.5 ns (the issue rate of plain f32 VALU at four waves per SIMD,
              profiles/r03_valu_rates.txt; 2 cycles at 2.4 GHz would be 0.83 ns -- the table prints both sums)
    ceiling = max of the three (perfect overlap of the three pipes inside the kernel, nothing else in the way)

and their sums: the attainable step, and -- for the kernels that make up the SRAttention + MLP blocks (everything but the heads, the MIM decoder, the embeddings, the
optimizer) -- the attainable blocks-only MFMA fraction north_star asks for (3 x 8.003 GFLOP x 256 pairs / sum of ceilings / 2.5 PFLOP/s).

    python3 tools/ceiling_table.py profiles/r06_step_sq.csv profiles/r06_step_traffic.txt [--md]      (--check: exit 1 if the two inputs were collected for different sources)
"""
import csv
import re
import sys

HBM_TBS, PEAK_PF, VALU_NS, VALU_NS_IDEAL = 6.3, 2.5, 1.5, 2.0 / 2.4
BLOCKS_FLOP = 3 * 8.003e9 * 256

# kernels of the heads / MIM decoder / embeddings / optimizer (everything else is counted to the blocks; the shared GEMM kernels are split by their share below)
NOT_BLOCKS = ("conv3_", "bn_", "col_reduce", "upsample", "ew_mul", "smooth_l1", "ce_", "adamw", "bert_embed", "patchify", "masked_select", "weight_prep", "fold_copies",
              "loss_compose", "head_grad", "scatter_rows", "gather_rows", "keep_mask", "droppath", "resize_tokens", "at::", "__amd", "tn_fold", "batch_sum", "add_column",
              "row_scale", "gelu_bwd", "cast", "grid_", "token_mask")


def short(n):
    n = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", n)
    return n


def load_sq(path):
    rows = list(csv.reader(open(path)))
    head = rows[0]
    m = re.search(r"source_hash=([0-9a-f]+)", head[0])
    idx = {c: i for i, c in enumerate(head)}
    out = {}
    for r in rows[1:]:
        out[r[0].strip()] = dict(calls=float(r[1]), us=float(r[2]), valu=float(r[idx["SQ_INSTS_VALU"]]), mops=float(r[idx["SQ_INSTS_VALU_MFMA_MOPS_BF16"]]))
    return out, (m.group(1) if m else None)


def load_traffic(path):
    lines = open(path).read().split("\n")
    m = re.search(r"source_hash=([0-9a-f]+)", lines[0])
    out = {}
    for l in lines[2:]:
        p = l.split(None, 6)
        if len(p) == 7:
            out[p[6].strip()] = float(p[0]) * 1e6
    return out, (m.group(1) if m else None)


def main():
    sq, h1 = load_sq(sys.argv[1])
    tr, h2 = load_traffic(sys.argv[2])
    md = "--md" in sys.argv
    if "--check" in sys.argv and h1 != h2:
        print(f"inputs collected for different kernel sources: {h1} / {h2}")
        sys.exit(1)
    rows = []
    for n, d in sq.items():
        byt = tr.get(n, 0.0)
        mfma_inst = d["mops"] / 32.0
        t_b = byt / (HBM_TBS * 1e12) * 1e6
        t_m = d["mops"] * 512.0 / (PEAK_PF * 1e15) * 1e6
        t_v = max(0.0, d["valu"] - mfma_inst) / 1024.0 * VALU_NS * 1e-3
        t_vi = max(0.0, d["valu"] - mfma_inst) / 1024.0 * VALU_NS_IDEAL * 1e-3
        rows.append(dict(n=short(n), calls=d["calls"], us=d["us"], tb=t_b, tm=t_m, tv=t_v, tvi=t_vi, ceil=max(t_b, t_m, t_v), ceil_i=max(t_b, t_m, t_vi),
                         blocks=not short(n).startswith(NOT_BLOCKS)))
    rows.sort(key=lambda r: -r["us"])
    tot = {k: sum(r[k] for r in rows) for k in ("us", "tb", "tm", "tv", "tvi", "ceil", "ceil_i")}
    blk = {k: sum(r[k] for r in rows if r["blocks"]) for k in ("us", "ceil", "ceil_i", "tm")}
    sep = " | " if md else "  "
    hdr = ["kernel", "launches", "measured us", "t_bytes", "t_mfma", "t_valu", "ceiling us", "limit", "measured / ceiling"]
    if md:
        print("| " + " | ".join(hdr) + " |")
        print("|" + "---|" * len(hdr))
    else:
        print(f"{'measured':>9} {'t_bytes':>8} {'t_mfma':>8} {'t_valu':>8} {'ceiling':>8} {'x':>5} {'calls':>6}  kernel")
    shown = 0.0
    for r in rows:
        if r["us"] < 60.0:
            continue
        shown += r["us"]
        lim = "HBM" if r["ceil"] == r["tb"] else ("MFMA" if r["ceil"] == r["tm"] else "VALU")
        x = r["us"] / max(r["ceil"], 1e-9)
        if md:
            print(f"| `{r['n'][:58]}` | {r['calls']:.0f} | {r['us']:.0f} | {r['tb']:.0f} | {r['tm']:.0f} | {r['tv']:.0f} | {r['ceil']:.0f} | {lim} | {x:.2f} |")
        else:
            print(f"{r['us']:9.1f} {r['tb']:8.1f} {r['tm']:8.1f} {r['tv']:8.1f} {r['ceil']:8.1f} {x:5.2f} {r['calls']:6.1f}  {r['n'][:80]} [{lim}]")
    rest = [r for r in rows if r["us"] < 60.0]
    if md:
        print(f"| the other {len(rest)} kernels (< 60 us per step each) | {sum(r['calls'] for r in rest):.0f} | {sum(r['us'] for r in rest):.0f} | {sum(r['tb'] for r in rest):.0f} | "
              f"{sum(r['tm'] for r in rest):.0f} | {sum(r['tv'] for r in rest):.0f} | {sum(r['ceil'] for r in rest):.0f} | | {sum(r['us'] for r in rest) / max(1e-9, sum(r['ceil'] for r in rest)):.2f} |")
        print(f"| **one step** | {sum(r['calls'] for r in rows):.0f} | **{tot['us']:.0f}** | {tot['tb']:.0f} | {tot['tm']:.0f} | {tot['tv']:.0f} | **{tot['ceil']:.0f}** | | {tot['us'] / tot['ceil']:.2f} |")
    print()
    print(f"step: measured {tot['us'] / 1e3:.2f} ms of kernel time (under the counter pass); bytes alone {tot['tb'] / 1e3:.2f} ms, MFMA alone {tot['tm'] / 1e3:.2f} ms, VALU alone {tot['tv'] / 1e3:.2f} ms "
          f"(at 1.5 ns per wave-instruction and SIMD; {tot['tvi'] / 1e3:.2f} ms at the 2-cycle issue of the data sheet); sum of per-kernel ceilings {tot['ceil'] / 1e3:.2f} ms "
          f"({tot['ceil_i'] / 1e3:.2f} ms with the data-sheet VALU rate) = {tot['us'] / tot['ceil']:.2f} x below the measured step.")
    print(f"blocks (every kernel but the heads / MIM decoder / embeddings / optimizer): measured {blk['us'] / 1e3:.2f} ms, ceilings {blk['ceil'] / 1e3:.2f} ms "
          f"({blk['ceil_i'] / 1e3:.2f}) -> attainable blocks-only MFMA fraction {BLOCKS_FLOP / (blk['ceil'] * 1e-6) / (PEAK_PF * 1e15):.3f} "
          f"({BLOCKS_FLOP / (blk['ceil_i'] * 1e-6) / (PEAK_PF * 1e15):.3f}), measured in this pass {BLOCKS_FLOP / (blk['us'] * 1e-6) / (PEAK_PF * 1e15):.3f}; "
          f"the blocks' executed MFMA time alone is {blk['tm'] / 1e3:.2f} ms.  [sources: sq {h1}, traffic {h2}]")


if __name__ == "__main__":
    main()
