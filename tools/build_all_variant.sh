#!/bin/bash
# an experimental build of the WHOLE library with extra -D flags (flags that live in the shared headers): tools/build_all_variant.sh NAME -DFLAG ... -> build_x/libsfa_NAME.so
set -e
D=$(cd "$(dirname "$0")/../slowflow_amd/csrc" && pwd)
N=$1; shift
mkdir -p $D/build_x/all_$N
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function"
for s in kernels sor sor_chain occlusion api; do
  X=""; [ $s = kernels ] && X="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc $F $X "$@" -c $D/$s.hip -o $D/build_x/all_$N/$s.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/build_x/libsfa_$N.so $D/build_x/all_$N/*.o
echo built $D/build_x/libsfa_$N.so
