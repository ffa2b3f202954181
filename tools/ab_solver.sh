#!/bin/bash
# A/B of solver builds inside one box: the solver alone per batch size and shape, then the whole path.  usage (GPU box): bash tools/ab_solver.sh NAME ... (build_x/libsfa_NAME.so; "default" = the product)
X=slowflow_amd/csrc/build_x
for n in "$@"; do
  if [ $n = default ]; then unset SFA_LIB; else export SFA_LIB=$X/libsfa_$n.so; fi
  echo "== $n"
  timeout -k 10 400 python3 tools/bench_sor_chain.py "8 16 32 64" "11 13 3" || exit 1
  timeout -k 10 300 python3 bench.py --batch 128 --streams 2 --steps 6 --warmup 2 --no-cpu-baseline --path-only | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n whole path 128 x 2:', round(d['value']), d['ms_per_step'], flush=True)" || exit 1
done
# the whole path with another solver shape for every launch: SHAPES="13 11" bash tools/ab_solver.sh ...
unset SFA_LIB
for sh in $SHAPES; do
  SFA_DEBUG=1 SFA_SOR_CHAIN=$sh timeout -k 10 300 python3 bench.py --batch 128 --streams 2 --steps 6 --warmup 2 --no-cpu-baseline --path-only | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default lib, shape $sh, whole path 128 x 2:', round(d['value']), d['ms_per_step'], flush=True)" || exit 1
done
