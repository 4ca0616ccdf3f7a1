#!/bin/bash
# Run on the GPU box (through gpurun): the measurement set of a round -- bench line, kernel-trace stats of the bench command, per-shape
# GEMM inventory, SQ counters of the hot kernels, HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the roofline launches.
#   gpurun --timeout 2400 -- 'bash tools/collect_round.sh r02'        then copy gpurun_out/<tag>_* into profiles/
set -u
tag=${1:-r02}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_n1.json 2> gpurun_out/${tag}_bench_n1.err
cut -c1-300 gpurun_out/${tag}_bench_n1.json
bash tools/profile_bench.sh $tag --steps 6 --warmup 3 > /dev/null 2>&1
cp gpurun_out/kernel_stats_$tag.csv gpurun_out/${tag}_kernel_stats.csv 2>/dev/null
python3 tools/step_launches.py gpurun_out/prof_$tag/trace_kernel_trace.csv > gpurun_out/${tag}_step_launches.txt 2>&1
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 MVLT_DP_FORCE_COLLECTIVES=1 python3 tools/overlap_trace.py > gpurun_out/${tag}_overlap.txt 2>/dev/null
bash tools/l1_stalls.sh > /dev/null 2>&1; cp gpurun_out/l1_stalls.txt gpurun_out/${tag}_l1_stalls.txt 2>/dev/null
bash tools/step_traffic.sh $tag > /dev/null 2>&1
bash tools/step_sq.sh $tag > /dev/null 2>&1                      # SQ instruction counters per kernel over whole steps (tools/ceiling_table.py)
python3 tools/gemm_shapes.py > gpurun_out/${tag}_gemm_shapes.txt 2>&1
(python3 tools/host_time.py; python3 tools/host_waits.py 20) 2>&1 | grep -E "^host|^pretrain:|^finetune:" > gpurun_out/${tag}_host_time.txt
bash tools/pmc_collect.sh $tag > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/pmc_${tag}_$c; rm -rf "$out"; mkdir -p "$out"
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out" -o p -- python3 tools/roofline_launch.py > "$out/stdout.txt" 2> "$out/stderr.txt"
  find "$out" -name "*kernel_trace.csv" -delete
done
python3 tools/roofline_traffic.py gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE gpurun_out/${tag}_roofline_traffic.json
cat gpurun_out/${tag}_roofline_traffic.json | head -c 600
ls gpurun_out | grep "^${tag}_"
