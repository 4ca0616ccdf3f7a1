#!/bin/bash
# A/B of env switches inside ONE gpurun call (box-to-box spread is 2-4 %): bash tools/ab_bench.sh "" "MVLT_ATTN_FWD_LEGACY=1" ...
# each argument is an environment assignment list (possibly empty) for one bench run; two passes to expose drift
for pass in 1 2; do
  for cfg in "$@"; do
    env $cfg python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-40s %9.1f pairs/s  %7.3f ms/step  blocks %7.3f ms (%.4f)' % ('[' + sys.argv[1] + ']', d['value'], d['ms_per_step'], d['flops']['blocks_only']['ms_per_step'], d['flops']['blocks_only']['mfma_frac']))" "$cfg"
  done
done
