#!/bin/bash
# k_warp_smooth: plain (x, y, window) tile order against the XCD-contiguous order (tools/build_variant.sh wsx -fno-slp-vectorize -DSFA_WS_XCD=1, SFA_VARIANT_SRC=kernels):
# launch duration (kernel trace) AND fabric-side bytes (FETCH_SIZE, WRITE_SIZE: separate --pmc passes), bench workload at 128 windows.  usage (GPU box): bash tools/ab_warp_smooth.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for n in default wsx default wsx; do
  if [ $n = default ]; then unset SFA_LIB; else export SFA_LIB=$R/slowflow_amd/csrc/build_x/libsfa_$n.so; fi
  d=$R/gpurun_out/ws_$n
  rm -rf $d; timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $d -o a -f csv -- python3 $R/tools/bench_kernels.py 128 > /dev/null 2>&1 || { echo "$n: trace failed"; exit 1; }
  echo "== $n: duration"; python3 $R/tools/profsum.py $(find $d -name "*kernel_stats.csv") 4 | grep -E "warp_smooth|assemble"
  rm -rf $d
done
for n in default wsx; do
  if [ $n = default ]; then unset SFA_LIB; else export SFA_LIB=$R/slowflow_amd/csrc/build_x/libsfa_$n.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    d=$R/gpurun_out/ws_${n}_$c
    rm -rf $d; timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c -d $d -o a -f csv -- python3 $R/tools/bench_kernels.py 128 > /dev/null 2>&1 || { echo "$n $c: pmc failed"; exit 1; }
    echo "== $n: $c (mean per dispatch, rocprofv3 units of 1024 B; reads count twice on gfx950)"; python3 $R/tools/pmc_sum.py $(find $d -name "*counter_collection.csv") warp_smooth
    rm -rf $d
  done
done
