#!/bin/bash
# Run on the GPU box (through gpurun): L2 hit / miss / fabric-read counters of the stage 3-4 MLP GEMM launches (tools/ubench_mlpgemm.py), one --pmc pass.
#   gpurun --timeout 900 -- 'bash tools/l2_hits.sh'   -> gpurun_out/l2_hits.txt
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/pmc_l2; rm -rf "$out"; mkdir -p "$out"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --kernel-trace --output-format csv -d "$out" -o p -- python3 tools/ubench_mlpgemm.py > "$out/stdout.txt" 2> "$out/stderr.txt"
python3 - "$out" > gpurun_out/l2_hits.txt <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for path in f:
    for r in csv.DictReader(open(path)):
        k = (r["Kernel_Name"][:70], r.get("Grid_Size", ""))
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
for k, c in sorted(acc.items()):
    calls = max(n[(k, name)] for name in c)
    hit, miss = c.get("TCC_HIT_sum", 0) / calls, c.get("TCC_MISS_sum", 0) / calls
    print(f"{k[0]:70s} grid {k[1]:>9s} calls {calls:3d}  L2 hit rate {hit / max(hit + miss, 1):.3f}  requests/launch {c.get('TCC_REQ_sum', 0) / calls:.3e}  fabric reads/launch {c.get('TCC_EA0_RDREQ_sum', 0) / calls:.3e} (x 64 B = {c.get('TCC_EA0_RDREQ_sum', 0) / calls * 64 / 1e6:.0f} MB as counted)")
PY
cat gpurun_out/l2_hits.txt
