set -e
mkdir -p gpurun_out/r4g
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout -k 10 400 python bench.py > gpurun_out/r4g/bench.json 2> gpurun_out/r4g/bench.err; echo "bench rc $?"; python - <<'P'
import json
d=json.load(open('gpurun_out/r4g/bench.json'))
print({k:d[k] for k in ('value','ms_per_step')})
print(json.dumps(d['roofline'],indent=0)[:3000])
print(json.dumps(d.get('roofline_assemble'),indent=0)[:1500])
print(d.get('config4_strong')); print(d.get('config5_strong')); print(d.get('latency_one_window_ms'), d.get('cfg_schedule_with_thresholds'))
P
