#!/bin/bash
# does the 4 KB row pitch of 1024-column frames cost the streaming kernels?  product against -DSFA_PITCH_ODD=1 (api.hip variant `podd`: 1088-column rows at level 0, 832 at level 3):
# per-level durations of every kernel of the bench workload (kernel trace), then the bench's timed step.  usage (GPU box): bash tools/pitch_whatif.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for n in default podd; do
  if [ $n = default ]; then unset SFA_LIB; else export SFA_LIB=$R/slowflow_amd/csrc/build_x/libsfa_$n.so; fi
  d=$R/gpurun_out/pw_$n
  rm -rf $d; timeout -k 5 150 rocprofv3 --kernel-trace -d $d -o a -f csv -- python3 $R/tools/bench_kernels.py 128 > /dev/null 2>&1 || { echo "$n: failed"; continue; }
  echo "== $n"; python3 $R/tools/by_level.py $(find $d -name "*kernel_trace.csv") k_warp_smooth k_assemble_images k_sor_chain k_update_outer_x k_pyr_down k_dpsis
  rm -rf $d
done
cd $R
bash tools/ab_libs.sh default podd
