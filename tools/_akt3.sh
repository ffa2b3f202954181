cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  d=$GRAFT_REPO_ROOT/gpurun_out/akt_ws$v
  if [ $v = 1 ]; then export SFA_NO_WARP_SMOOTH=1; else unset SFA_NO_WARP_SMOOTH; fi
  timeout -k 5 60 rocprofv3 --kernel-trace --stats -d $d -o a -f csv -- python3 $GRAFT_REPO_ROOT/tools/bench_kernels.py 64 > /dev/null 2>&1 || { echo FAILED; exit 1; }
  echo "== NO_WARP_SMOOTH=$v"; python3 $GRAFT_REPO_ROOT/tools/profsum.py $(find $d -name "*kernel_stats.csv") 6
  rm -rf $d
done
unset SFA_NO_WARP_SMOOTH
cd $GRAFT_REPO_ROOT && timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
