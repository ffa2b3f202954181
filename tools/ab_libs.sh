#!/bin/bash
# A/B of whole libraries inside ONE box (boxes differ by 3-5 %): the bench's timed step and the one-window latency per library, the list twice.
# usage (GPU box): bash tools/ab_libs.sh NAME ...   (build_x/libsfa_NAME.so; "default" = the product)
X=$GRAFT_REPO_ROOT/slowflow_amd/csrc/build_x
for rep in 1 2; do for n in "$@"; do
  if [ $n = default ]; then unset SFA_LIB; else export SFA_LIB=$X/libsfa_$n.so; fi
  timeout -k 10 300 python3 bench.py --bench-only --steps 8 --warmup 2 --no-cpu-baseline | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n bench', round(d['value']), d['ms_per_step'], flush=True)" || exit 1
  timeout -k 10 120 python3 tools/bench_one_window.py 3 | tail -1 || exit 1
done; done
