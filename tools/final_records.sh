# the round's final records in ONE box (boxes differ by 3-5 %): run from the repo root on the GPU box;  bash tools/final_records.sh [bench-only]
set -e
mkdir -p gpurun_out/r04_summary
if [ "$1" != "bench-only" ]; then
  bash profiles/collect.sh r04 128 > gpurun_out/r04_collect.log 2>&1
  python3 profiles/summarize.py r04 64 > gpurun_out/r04_summarize.log 2>&1
  cp profiles/r04_* gpurun_out/r04_summary/
  rm -rf gpurun_out/r04_sq1 gpurun_out/r04_sq2 gpurun_out/r04_fetch gpurun_out/r04_write gpurun_out/r04_stats gpurun_out/r04_stats1
fi
python3 bench.py > gpurun_out/r04_summary/r04_bench.json 2> gpurun_out/r04_summary/r04_bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_summary/r04_bench_steps20.json 2>> gpurun_out/r04_summary/r04_bench.err
echo collected
