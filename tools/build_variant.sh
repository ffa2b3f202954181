#!/bin/bash
# experimental build of the library with extra -D flags for sor_chain.hip (timing / what-if experiments): tools/build_variant.sh NAME -DFLAG ...
# -> slowflow_amd/csrc/build_x/libsfa_NAME.so (use with SFA_LIB=...)
set -e
D=$(cd "$(dirname "$0")/../slowflow_amd/csrc" && pwd)
N=$1; shift
mkdir -p $D/build_x
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function"
SRC=${SFA_VARIANT_SRC:-sor_chain}
[ $SRC = kernels ] && F="$F -fno-slp-vectorize"      # as the Makefile does (SLP packing costs the data-term kernel 30 %)
/opt/rocm/bin/hipcc $F "$@" -c $D/$SRC.hip -o $D/build_x/${SRC}_$N.o
OBJS=""
for o in kernels sor sor_chain occlusion api; do if [ $o = $SRC ]; then OBJS="$OBJS $D/build_x/${SRC}_$N.o"; else OBJS="$OBJS $D/$o.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/build_x/libsfa_$N.so $OBJS
echo built $D/build_x/libsfa_$N.so
