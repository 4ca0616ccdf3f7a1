#!/usr/bin/env python3
"""Scan the gfx950 ISA of every kernel for an s_barrier that a loop back-edge reaches while LDS writes of the same wave may still
be in flight (no `s_waitcnt ... lgkmcnt(0)` between the last ds_write / LDS atomic before the branch and the barrier at the loop
top).  This is the pattern behind the attention-backward race of round 1: hipcc dropped its own lgkmcnt(0) next to a hand-written
`s_waitcnt vmcnt(0)`.  Runs on the CPU (hipcc -S):   python tools/isa_barrier_check.py"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mvlt_amd", "csrc")


def functions(path):
    cur, out = None, {}
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1); out[cur] = []
        if cur:
            out[cur].append(line)
        if "s_endpgm" in line:
            cur = None
    return out


def scan(fn):
    labels = {m.group(1): i for i, l in enumerate(fn) for m in [re.match(r"^(\.LBB\w+):", l)] if m}
    hits = []
    for i, l in enumerate(fn):
        m = re.match(r"\s*s_c?branch\w*\s+(\.LBB\w+)", l)
        if not m or labels.get(m.group(1), i) >= i:
            continue
        j, unguarded = labels[m.group(1)], False
        while j < i:
            t = fn[j].strip()
            if "lgkmcnt(0)" in t or t.startswith(("s_cbranch", "s_branch")):
                break
            if t == "s_barrier":
                unguarded = True; break
            j += 1
        if not unguarded:
            continue
        k = i - 1
        while k > labels[m.group(1)]:
            t = fn[k].strip()
            if "lgkmcnt(0)" in t:
                break
            if t.startswith(("ds_write", "ds_add", "ds_max", "ds_min")):
                hits.append((i, j, k)); break
            k -= 1
    return hits


def check_file(args):
    f, tmp = args
    asm = os.path.join(tmp, f + ".s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-S", "--cuda-device-only",
                    os.path.join(SRC, f), "-o", asm], check=True, stderr=subprocess.DEVNULL)
    out = []
    for name, fn in functions(asm).items():
        for i, j, k in scan(fn):
            out.append(f"{f}: {name[:90]}: back-edge at +{i} reaches the barrier at +{j} with the LDS write at +{k} undrained")
    return out


def main():
    from concurrent.futures import ThreadPoolExecutor
    files = sorted((f for f in os.listdir(SRC) if f.endswith(".hip")), key=lambda f: -os.path.getsize(os.path.join(SRC, f)))
    with tempfile.TemporaryDirectory() as tmp, ThreadPoolExecutor(max_workers=min(6, len(files))) as ex:     # one hipcc -S per source, in parallel
        hits = [h for r in ex.map(check_file, [(f, tmp) for f in files]) for h in r]
    for h in hits:
        print(h)
    print("suspicious back-edges:", len(hits))
    return 1 if hits else 0


if __name__ == "__main__":
    sys.exit(main())
