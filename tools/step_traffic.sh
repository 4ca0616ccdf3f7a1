#!/bin/bash
# gpurun --timeout 1500 -- 'bash tools/step_traffic.sh r02'  -> gpurun_out/<tag>_step_traffic.txt
set -u
tag=${1:-r02}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/pmcstep_${tag}_$c; rm -rf "$out"; mkdir -p "$out"
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out" -o p -- python3 bench.py --steps 4 --warmup 0 --no-cpu-baseline --no-roofline > "$out/stdout.txt" 2> "$out/stderr.txt"
  find "$out" -name "*kernel_trace.csv" -delete
done
python3 tools/step_traffic.py gpurun_out/pmcstep_${tag}_FETCH_SIZE gpurun_out/pmcstep_${tag}_WRITE_SIZE 4 > gpurun_out/${tag}_step_traffic.txt
head -60 gpurun_out/${tag}_step_traffic.txt
rm -rf gpurun_out/pmcstep_${tag}_FETCH_SIZE gpurun_out/pmcstep_${tag}_WRITE_SIZE
