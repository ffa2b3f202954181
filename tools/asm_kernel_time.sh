#!/bin/bash
# usage (on the GPU box): bash tools/asm_kernel_time.sh <lib.so>...   -> average duration of the assembly and SOR kernels per library (rocprofv3 kernel trace, batch 64)
# every library runs under its own timeout: a what-if build computes garbage by construction and may feed the solver operands it never terminates on
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  d=$GRAFT_REPO_ROOT/gpurun_out/akt_$(basename $lib .so)
  SFA_LIB=$lib timeout -k 5 60 rocprofv3 --kernel-trace --stats -d $d -o a -f csv -- python3 $GRAFT_REPO_ROOT/tools/bench_kernels.py 64 > /dev/null 2>&1 || { echo "== $(basename $lib): FAILED or timed out"; exit 1; }
  echo "== $(basename $lib)"; python3 $GRAFT_REPO_ROOT/tools/profsum.py $(find $d -name "*kernel_stats.csv") 2
  rm -rf $d
done
