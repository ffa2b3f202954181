#!/bin/bash
# usage (GPU box): bash tools/bench_batches.sh "<batch>x<streams> ..."   e.g. "128x2 146x2 102x2"   -> value and ms/step per setting (path only, 5 steps)
for bs in $1; do
  b=${bs%x*}; s=${bs#*x}
  timeout -k 10 300 python3 bench.py --batch $b --streams $s --steps 5 --warmup 1 --no-cpu-baseline --path-only | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$bs', round(d['value']), d['ms_per_step'])" || exit 1
done
