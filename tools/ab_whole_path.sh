# A/B of whole-path throughput inside ONE box (boxes differ by 3-5 %): bash tools/ab_whole_path.sh "ENV1=.. ENV2=.." -> default first, then each setting, then default again
run() { timeout -k 10 300 python3 bench.py --batch 128 --streams 2 --steps 6 --warmup 2 --no-cpu-baseline --path-only | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), d['ms_per_step'])" || exit 1; }
export SFA_DEBUG=1   # the library reads its switches from the environment only behind this
run default
for e in "$@"; do ( export $e; run "$e" ) || exit 1; done
export SFA_DEBUG=1   # the library reads its switches from the environment only behind this
run default
