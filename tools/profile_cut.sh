#!/bin/bash
# where the cfg schedule's time goes at 16 windows per launch (one GPU's share of BASELINE config 4 on eight): kernel trace of tools/bench_cfg4.py -> tools/cut_profile.py
# usage (GPU box): bash tools/profile_cut.sh [tag] > gpurun_out/<tag>_cut_profile.txt
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/${TAG}_cutprof
rocprofv3 --kernel-trace -d gpurun_out/${TAG}_cutprof -o t -f rocpd -- python3 tools/bench_cfg4.py 16 > gpurun_out/${TAG}_cutprof.log 2>&1
grep -v amdgpu.ids gpurun_out/${TAG}_cutprof.log | tail -4
DB=$(find gpurun_out/${TAG}_cutprof -name "*.db" | head -1)
python3 tools/cut_profile.py "$DB"
python3 tools/trace_gaps_db.py "$DB"
rm -rf gpurun_out/${TAG}_cutprof
