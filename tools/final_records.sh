# the round's final records in ONE box (boxes differ by 3-5 %): run from the repo root on the GPU box;  bash tools/final_records.sh TAG HEAD [bench-only]
# HEAD = `git rev-parse --short HEAD` of the tree sent to the box (the box has no .git): it goes into every summary (profiles/summarize.py provenance)
set -e
TAG=${1:-r05}; HEAD=${2:-unknown}
mkdir -p gpurun_out/${TAG}_summary
if [ "$3" != "bench-only" ]; then
  bash profiles/collect.sh $TAG 128 $HEAD > gpurun_out/${TAG}_collect.log 2>&1
  python3 profiles/summarize.py $TAG 128 > gpurun_out/${TAG}_summarize.log 2>&1
  cp profiles/${TAG}_* gpurun_out/${TAG}_summary/
  rm -rf gpurun_out/${TAG}_sq1 gpurun_out/${TAG}_sq2 gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_stats gpurun_out/${TAG}_stats1
fi
python3 bench.py > gpurun_out/${TAG}_summary/${TAG}_bench.json 2> gpurun_out/${TAG}_summary/${TAG}_bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_summary/${TAG}_bench_steps20.json 2>> gpurun_out/${TAG}_summary/${TAG}_bench.err
echo "$HEAD" > gpurun_out/${TAG}_summary/${TAG}_bench_head.txt
echo collected
