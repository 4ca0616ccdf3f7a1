#!/bin/bash
# Run on the GPU box (through gpurun): texture-address / L1 (TA, TCP) busy, stall and latency counters of the stage 3-4 MLP GEMM launches
# (tools/ubench_mlpgemm.py), five small --pmc passes.   gpurun --timeout 900 -- 'bash tools/l1_stalls.sh'   -> gpurun_out/l1_stalls.txt
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
i=0
# (six counters of these blocks in one pass made rocprofv3 abort and then hang in its finaliser: small sets, each under its own timeout)
for set in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum"; do
  i=$((i+1)); out=gpurun_out/pmc_l1_$i; rm -rf "$out"; mkdir -p "$out"
  timeout -s KILL 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out" -o p -- python3 tools/ubench_mlpgemm.py > "$out/stdout.txt" 2> "$out/stderr.txt"
  echo "pass $i rc=$?"
done
python3 - > gpurun_out/l1_stalls.txt <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for path in glob.glob("gpurun_out/pmc_l1_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "gemm_" not in r["Kernel_Name"]:
            continue
        k = (r["Kernel_Name"].split("::")[-1][:48], r.get("Grid_Size", ""))
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
names = sorted({c for v in acc.values() for c in v})
for k, c in sorted(acc.items()):
    per = {name: c[name] / max(n[(k, name)], 1) for name in names}
    line = f"{k[0]:48s} grid {k[1]:>8s}  " + "  ".join(f"{name.replace('_sum', '')}={per[name]:.3e}" for name in names)
    if per.get("TCP_TCC_READ_REQ_sum"):
        line += f"   avg L1->L2 read latency {per['TCP_TCC_READ_REQ_LATENCY_sum'] / per['TCP_TCC_READ_REQ_sum']:.0f} clk"
    print(line)
PY
cat gpurun_out/l1_stalls.txt | cut -c1-700
