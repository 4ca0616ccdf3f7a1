#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace + stats of the bench command, summary copied to gpurun_out/.
#   gpurun --timeout 1200 -- 'bash tools/profile_bench.sh r01 --steps 6 --warmup 3'
# (the headline workload only: the short runs of BASELINE configurations #4 / #5 that the default bench line appends would mix other shapes into the
#  per-kernel averages)
set -u
tag=${1:-r01}; shift
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o trace -- python3 bench.py --gpus 1 --no-cpu-baseline --no-other-configs "$@" > "$out/bench_stdout.txt" 2> "$out/bench_stderr.txt"
echo "rocprofv3 rc=$?" >> "$out/bench_stdout.txt"
# keep only the small CSV summaries (the full trace can be large)
find "$out" -name "*kernel_stats.csv" -exec cp {} gpurun_out/kernel_stats_$tag.csv \;
find "$out" -name "*kernel_trace.csv" -size +40M -delete
tail -3 "$out/bench_stdout.txt"
head -25 gpurun_out/kernel_stats_$tag.csv 2>/dev/null | cut -c1-200
