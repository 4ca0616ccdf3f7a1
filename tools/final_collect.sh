#!/bin/bash
# The round's final collection in ONE gpurun call: the GPU suite (parity report), tools/collect_round.sh, then the bench line again with this collection's
# counter figures in place.   gpurun --timeout 3000 -- 'bash tools/final_collect.sh r05'
set -u
tag=${1:-r05}
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
python3 -m pytest tests -q -m gpu 2>&1 | tail -5 > gpurun_out/${tag}_gputest_tail.txt
python3 tools/parity_report.py gpurun_out/parity_report.json > gpurun_out/${tag}_parity_report.txt 2>&1
bash tools/collect_round.sh $tag > gpurun_out/collect_${tag}.log 2>&1
cp gpurun_out/${tag}_roofline_traffic.json gpurun_out/${tag}_step_traffic.txt gpurun_out/${tag}_step_sq.csv gpurun_out/${tag}_kernel_stats.csv profiles/ 2>/dev/null
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_n1.json 2> gpurun_out/${tag}_bench_n1.err
python3 tools/ft_ramp.py finetune 45 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_ft_ramp.txt
python3 tools/host_profile.py 2>&1 | grep -v amdgpu.ids | head -60 | cut -c1-170 > gpurun_out/${tag}_host_profile.txt
cut -c1-400 gpurun_out/${tag}_bench_n1.json
bash tools/profile_other_configs.sh > gpurun_out/${tag}_other_configs.log 2>&1      # kernel traces / per-shape inventory of BASELINE configurations #4 and #5
MODEL=pvlt_medium IMG=384 B=64 python3 tools/host_time.py 2>/dev/null | head -1 > gpurun_out/${tag}_medium384_host_time.txt
