#!/bin/bash
# same-box A/B of whole-step time for alternate builds: bash tools/ab_bench_libs.sh "" ab/libmvlt_X.so ...   ("" = the normal build); two passes
for pass in 1 2; do
  for lib in "$@"; do
    MVLT_HIP_LIB=$lib python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-32s %9.1f pairs/s  %7.3f ms/step  blocks %7.3f ms (%.4f)' % ('[' + sys.argv[1] + ']', d['value'], d['ms_per_step'], d['flops']['blocks_only']['ms_per_step'], d['flops']['blocks_only']['mfma_frac']))" "${lib:-default}"
  done
done
