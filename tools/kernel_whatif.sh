#!/bin/bash
# kernel-trace durations of the bench workload's kernels (128 windows) for the product and what-if builds: bash tools/kernel_whatif.sh NAME ...  (build_x/libsfa_NAME.so; "default" = the product)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  if [ $n = default ]; then unset SFA_LIB; else export SFA_LIB=$R/slowflow_amd/csrc/build_x/libsfa_$n.so; fi
  d=$R/gpurun_out/kw_$n
  rm -rf $d; timeout -k 5 120 rocprofv3 --kernel-trace --stats -d $d -o a -f csv -- python3 $R/tools/bench_kernels.py 128 > /dev/null 2>&1 || { echo "$n: failed or timed out"; rm -rf $d; continue; }
  echo "== $n"; python3 $R/tools/profsum.py $(find $d -name "*kernel_stats.csv") 5
  rm -rf $d
done
