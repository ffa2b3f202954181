#!/bin/bash
# what clock and power the GPU runs at under the bench workload: rocm-smi sampled every 0.5 s beside `bench.py --bench-only` (read-only queries; an ordinary user may read them)
# usage (GPU box): bash tools/clock_power_probe.sh
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction|memory)" | tr '\n' ' ' | sed 's/  */ /g'; echo; sleep 0.5; done ) > gpurun_out/clock_power_samples.txt &
S=$!
sleep 3
timeout -k 10 200 python3 bench.py --bench-only --steps 60 --warmup 2 --no-cpu-baseline | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(d['value']), d['ms_per_step'], flush=True)"
wait $S
awk 'NR%4==1' gpurun_out/clock_power_samples.txt | cut -c1-260
