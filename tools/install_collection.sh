#!/bin/bash
# After `gpurun -- 'bash tools/final_collect.sh rNN'`: copy the merged-back summaries from gpurun_out/ into profiles/ and regenerate the measured tables of DESIGN.md.
#   bash tools/install_collection.sh r06
set -e
tag=${1:-r06}
cd "$(dirname "$0")/.."
for f in bench_n1.json gemm_shapes.txt gputest_tail.txt host_profile.txt host_time.txt kernel_stats.csv l1_stalls.txt medium384_gemm_shapes.txt medium384_host_time.txt medium384_step_launches.txt \
         finetune_step_launches.txt mfma_counters.csv overlap.txt parity_report.txt roofline_traffic.json step_launches.txt step_sq.csv step_traffic.txt ft_ramp.txt smoke.txt repeat_step.txt; do
  [ -f gpurun_out/${tag}_$f ] && cp gpurun_out/${tag}_$f profiles/
done
[ -f gpurun_out/kernel_stats_${tag}_medium384.csv ] && cp gpurun_out/kernel_stats_${tag}_medium384.csv profiles/${tag}_medium384_kernel_stats.csv
[ -f gpurun_out/kernel_stats_${tag}_finetune.csv ] && cp gpurun_out/kernel_stats_${tag}_finetune.csv profiles/${tag}_finetune_kernel_stats.csv
python3 tools/design_tables.py $tag | tail -1
python3 - <<PY
import json
d = json.loads(open('profiles/${tag}_bench_n1.json').read().strip().split('\n')[-1])
r = d['roofline']
print('bench:', d['value'], 'pairs/s', d['ms_per_step'], 'ms; blocks-only', d['flops']['blocks_only']['mfma_frac'], '; HBM', d['step']['hbm_gb_per_step'], 'GB; roofline frac', r['frac'], 'in step', (r.get('frac_in_step') or {}).get('frac'), 'traffic', r['traffic'])
print('other:', {k: (v.get('pairs_s'), v.get('ms_per_step', v.get('ms_per_batch'))) for k, v in d['other_configs'].items()}, 'cpu', d['cpu_baseline']['value'])
PY
tail -1 profiles/${tag}_gputest_tail.txt
