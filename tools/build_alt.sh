#!/bin/bash
# An alternate build of libmvlt_hip.so for same-box A/B runs: bash tools/build_alt.sh NAME SRC.hip[,SRC2.hip...] [extra hipcc flags]
# recompiles the listed csrc/ sources with the extra flags (typically -DMVLT_...=...) and links them with the other objects of the
# normal build into ab/libmvlt_NAME.so (ab/ is git-ignored but travels with gpurun); select it with MVLT_HIP_LIB=ab/libmvlt_NAME.so.
set -e
cd "$(dirname "$0")/.."
name=$1; srcs=$2; shift 2
python -m mvlt_amd.build >/dev/null
mkdir -p ab
objs=$(ls mvlt_amd/csrc/_obj/*.o)
alt=""
for src in ${srcs//,/ }; do
  extra=""
  [ "$src" = "mlp.hip" ] && extra="-fno-slp-vectorize"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -w $extra "$@" -c mvlt_amd/csrc/$src -o ab/${name}_${src%.hip}.o &
  objs=$(echo "$objs" | grep -v "/${src%.hip}.o")
  alt="$alt ab/${name}_${src%.hip}.o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $objs $alt -o ab/libmvlt_$name.so
echo ab/libmvlt_$name.so
