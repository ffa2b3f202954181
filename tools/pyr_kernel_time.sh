#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  d=$GRAFT_REPO_ROOT/gpurun_out/akt_$(basename $lib .so)
  SFA_LIB=$lib rocprofv3 --kernel-trace --stats -d $d -o a -f csv -- python3 $GRAFT_REPO_ROOT/tools/bench_kernels.py 64 > /dev/null 2>&1
  echo "== $(basename $lib)"; python3 $GRAFT_REPO_ROOT/tools/profsum.py $(find $d -name "*kernel_stats.csv") 6 | grep pyr
done
