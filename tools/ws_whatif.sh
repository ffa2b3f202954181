#!/bin/bash
# what bounds k_warp_smooth: kernel-trace duration of the product against the timing-only what-if builds (tools/build_variant.sh wsN -fno-slp-vectorize -DSFA_X_WS=N, SFA_VARIANT_SRC=kernels)
# usage (GPU box): bash tools/ws_whatif.sh NAME ...   (build_x/libsfa_NAME.so; "default" = the product).  A what-if build feeds the later kernels garbage: each runs under its own timeout
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  if [ $n = default ]; then unset SFA_LIB; else export SFA_LIB=$R/slowflow_amd/csrc/build_x/libsfa_$n.so; fi
  d=$R/gpurun_out/wsw_$n
  rm -rf $d; timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $d -o a -f csv -- python3 $R/tools/bench_kernels.py 128 > /dev/null 2>&1 || { echo "$n: failed or timed out"; rm -rf $d; continue; }
  echo "== $n"; python3 $R/tools/profsum.py $(find $d -name "*kernel_stats.csv") 6 | grep -E "warp_smooth|update_outer"
  rm -rf $d
done
