#!/bin/bash
# same-box A/B of alternate builds (tools/build_alt.sh): bash tools/ab_libs.sh SCRIPT.py "" ab/libmvlt_X.so ...   ("" = the normal build)
script=$1; shift
for lib in "$@"; do
  echo "=== lib: ${lib:-default}"
  MVLT_HIP_LIB=$lib python3 $script 2>&1 | grep -v amdgpu.ids
done
