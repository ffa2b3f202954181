set -e
bash profiles/collect.sh r04 128 > gpurun_out/r04_collect.log 2>&1 || { tail -5 gpurun_out/r04_collect.log; exit 1; }
python3 profiles/summarize.py r04 64 > gpurun_out/r04_summarize.log 2>&1 || { tail -5 gpurun_out/r04_summarize.log; exit 1; }
mkdir -p gpurun_out/r04_summary; cp profiles/r04_* gpurun_out/r04_summary/
cp gpurun_out/r04_stats.json gpurun_out/r04_summary/ || true
rm -rf gpurun_out/r04_sq1 gpurun_out/r04_sq2 gpurun_out/r04_fetch gpurun_out/r04_write gpurun_out/r04_stats gpurun_out/r04_stats1
tail -30 gpurun_out/r04_summarize.log
