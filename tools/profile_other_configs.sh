#!/bin/bash
# kernel trace / per-shape inventory of BASELINE configurations #4 (pvlt_medium, 384 px, batch 64) and #5 (fine-tune)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06a_bench_n1.json 2> gpurun_out/r06a_bench_n1.err
cut -c1-200 gpurun_out/r06a_bench_n1.json
bash tools/profile_bench.sh r06_medium384 --model pvlt_medium --img 384 --batch 64 --steps 4 --warmup 2 --no-roofline > /dev/null 2>&1
python3 tools/step_launches.py gpurun_out/prof_r06_medium384/trace_kernel_trace.csv > gpurun_out/r06_medium384_step_launches.txt 2>&1
head -40 gpurun_out/r06_medium384_step_launches.txt
MODEL=pvlt_medium IMG=384 B=64 python3 tools/gemm_shapes.py > gpurun_out/r06_medium384_gemm_shapes.txt 2>&1
head -50 gpurun_out/r06_medium384_gemm_shapes.txt
bash tools/profile_bench.sh r06_finetune --task finetune --steps 6 --warmup 3 --no-roofline > /dev/null 2>&1
python3 tools/step_launches.py gpurun_out/prof_r06_finetune/trace_kernel_trace.csv > gpurun_out/r06_finetune_step_launches.txt 2>&1
head -30 gpurun_out/r06_finetune_step_launches.txt
rm -rf gpurun_out/prof_r06_medium384/*trace.csv gpurun_out/prof_r06_finetune/*trace.csv
