#!/bin/bash
# A/B of env switches on another bench configuration inside ONE gpurun call, three interleaved passes:
#   bash tools/ab_configs.sh "--task finetune --steps 20 --warmup 5" "" "MVLT_ATTN_FWD_LEGACY=1" ...
args=$1; shift
for pass in 1 2 3; do
  for cfg in "$@"; do
    env $cfg python3 bench.py $args --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-44s %9.1f pairs/s %8.3f ms/step' % ('[' + sys.argv[1] + ']', d['value'], d['ms_per_step']))" "$cfg"
  done
done
