#!/bin/bash
# Does a HIP runtime setting change what the path costs?  One frame window alone (launch-latency-bound: ~175 launches), the cfg schedule at 16 windows (the cut's
# ~60 small launches per call) and the bench's timed step, each under the runtime's defaults and under the settings below.  The library never sets any of them: a
# setting that pays would be a line for INTEGRATION.md (the caller's environment), not a code change.   usage (GPU box): bash tools/runtime_env_probe.sh
R=$GRAFT_REPO_ROOT
run() {
  echo "== $1"
  timeout -k 10 120 python3 $R/tools/bench_one_window.py 4 2>&1 | grep "^run" | tail -2 | tr "\n" " "; echo
  timeout -k 10 200 python3 $R/tools/bench_cfg4.py 16 1 2>&1 | grep "per run"
  timeout -k 10 300 python3 $R/bench.py --bench-only --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(d['value']), d['ms_per_step'], flush=True)"
}
run "defaults"
HIP_FORCE_DEV_KERNARG=1 run "HIP_FORCE_DEV_KERNARG=1"
HIP_FORCE_DEV_KERNARG=0 run "HIP_FORCE_DEV_KERNARG=0"
GPU_MAX_HW_QUEUES=1 run "GPU_MAX_HW_QUEUES=1"
AMD_SERIALIZE_KERNEL=0 HSA_ENABLE_INTERRUPT=0 run "HSA_ENABLE_INTERRUPT=0 (busy-wait for completion signals)"
run "defaults again"
