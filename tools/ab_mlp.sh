#!/bin/bash
# A/B of library builds on one box: fused-MLP microbenchmark for every ab/*.so (and optionally the whole step)
for lib in ab/*.so; do
  echo "== $lib"
  MVLT_HIP_LIB=$PWD/$lib python tools/ubench_mlp.py 2>/dev/null
done
if [ "${1:-}" = "step" ]; then
for lib in ab/*.so; do
  echo "== $lib"
  MVLT_HIP_LIB=$PWD/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-180
done
fi
