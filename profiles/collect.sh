#!/bin/bash
# Profiles of the bench command, run on the GPU box (gpurun) from the repo root:
#   bash profiles/collect.sh r01        -> gpurun_out/<tag>_{stats,fetch,write}/...
# pass 1: kernel trace + stats; pass 2/3: HBM-side PMC counters, each in its own run (MI355X_MICROARCH.md:
# FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2; --pmc is never combined with the trace domains).
# Afterwards, in the container:  python profiles/summarize.py <tag> <windows per launch = batch / streams>   (writes profiles/<tag>_*.{csv,json})
set -e
TAG=${1:-r01}
BATCH=${2:-64}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_stats -o bench -f csv -- python3 bench.py --batch $BATCH --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_stats.json 2> gpurun_out/${TAG}_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${TAG}_fetch -o bench -f csv -- python3 bench.py --batch $BATCH --steps 2 --warmup 1 --no-cpu-baseline --path-only > gpurun_out/${TAG}_fetch.json 2> gpurun_out/${TAG}_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${TAG}_write -o bench -f csv -- python3 bench.py --batch $BATCH --steps 2 --warmup 1 --no-cpu-baseline --path-only > gpurun_out/${TAG}_write.json 2> gpurun_out/${TAG}_write.err
echo done
