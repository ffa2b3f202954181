#!/bin/bash
# Profiles of the bench command, run on the GPU box (gpurun) from the repo root:
#   bash profiles/collect.sh r02 [batch] [head]  -> gpurun_out/<tag>_{stats,fetch,write,sq1,sq2}/...
# head: `git rev-parse --short HEAD` of the tree the box was sent (the box has no .git: pass it from the container, e.g. gpurun -- "bash profiles/collect.sh r05 128 $(git rev-parse --short HEAD)");
# it travels in gpurun_out/<tag>_head.txt and profiles/summarize.py refuses to summarise under another HEAD (round 4's tables were two commits behind)
# pass 1: kernel trace + stats; passes 2/3: HBM-side PMC counters; passes 4/5: SQ counters (VALU / LDS activity, instruction mix) -- every --pmc pass in its
# own run with --kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2; --pmc is never combined with the trace domains).
# Afterwards, in the container:  python profiles/summarize.py <tag> <windows per launch = batch / streams>   (writes profiles/<tag>_*.{csv,json})
set -e
TAG=${1:-r03}
BATCH=${2:-128}
HEAD=${3:-unknown}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "$HEAD" > gpurun_out/${TAG}_head.txt
sha256sum slowflow_amd/libslowflow_amd.so | cut -d" " -f1 >> gpurun_out/${TAG}_head.txt
# pass 1: --bench-only: every launch of the process belongs to the timed workload, so the per-kernel averages are the bench's launches
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_stats -o bench -f csv -- python3 bench.py --batch $BATCH --steps 5 --warmup 1 --no-cpu-baseline --no-strong --bench-only > gpurun_out/${TAG}_stats.json 2> gpurun_out/${TAG}_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${TAG}_fetch -o bench -f csv -- python3 bench.py --batch $BATCH --steps 2 --warmup 1 --no-cpu-baseline --bench-only > gpurun_out/${TAG}_fetch.json 2> gpurun_out/${TAG}_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${TAG}_write -o bench -f csv -- python3 bench.py --batch $BATCH --steps 2 --warmup 1 --no-cpu-baseline --bench-only > gpurun_out/${TAG}_write.json 2> gpurun_out/${TAG}_write.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d gpurun_out/${TAG}_sq1 -o bench -f csv -- python3 bench.py --batch $BATCH --steps 2 --warmup 1 --no-cpu-baseline --bench-only > gpurun_out/${TAG}_sq1.json 2> gpurun_out/${TAG}_sq1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d gpurun_out/${TAG}_sq2 -o bench -f csv -- python3 bench.py --batch $BATCH --steps 2 --warmup 1 --no-cpu-baseline --bench-only > gpurun_out/${TAG}_sq2.json 2> gpurun_out/${TAG}_sq2.err
# pass 6: the same workload on ONE stream (no second group sharing the GPU): the per-kernel maxima here against pass 1 show what the overlap costs a kernel
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_stats1 -o bench -f csv -- python3 bench.py --batch $((BATCH / 2)) --streams 1 --steps 5 --warmup 1 --no-cpu-baseline --bench-only > gpurun_out/${TAG}_stats1.json 2> gpurun_out/${TAG}_stats1.err
echo done
# The raw counter files of the six passes are ~70 MB, more than gpurun merges back: summarise on the box and keep the summaries, e.g.
#   bash profiles/collect.sh r02 128; python3 profiles/summarize.py r02 64; mkdir -p gpurun_out/r02_summary; cp profiles/r02_* gpurun_out/r02_summary/;
#   rm -rf gpurun_out/r02_{sq1,sq2,fetch,write,stats,stats1}
