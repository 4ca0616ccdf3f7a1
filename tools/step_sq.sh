#!/bin/bash
# gpurun --timeout 1500 -- 'bash tools/step_sq.sh r06'  -> gpurun_out/<tag>_step_sq.csv: SQ instruction counters per kernel over whole steps of the bench process
# (one --pmc pass with --kernel-trace only; joined with <tag>_step_traffic.txt by tools/ceiling_table.py)
set -u
tag=${1:-r06}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/pmcstep_${tag}_SQ; rm -rf "$out"; mkdir -p "$out"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$out" -o p -- python3 bench.py --steps 4 --warmup 0 --no-cpu-baseline --no-roofline --no-other-configs > "$out/stdout.txt" 2> "$out/stderr.txt"
python3 tools/step_sq.py "$out" > gpurun_out/${tag}_step_sq.csv
head -5 gpurun_out/${tag}_step_sq.csv
rm -rf "$out"
