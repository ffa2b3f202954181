#!/bin/bash
# usage (GPU box): bash tools/pmc_kernels.sh <tag> <counter> [<counter>...]   -> gpurun_out/pmc_<tag>/  (one --pmc pass over tools/bench_kernels.py, batch 64)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o a -f csv -- python3 $GRAFT_REPO_ROOT/tools/bench_kernels.py 64 > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
