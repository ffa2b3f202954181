set -e
bash tools/asm_kernel_time.sh $PWD/slowflow_amd/libslowflow_amd.so 2>&1
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
timeout -k 10 300 python bench.py --no-cpu-baseline --no-strong > gpurun_out/r4c/bench2.json 2> gpurun_out/r4c/bench2.err && cut -c1-300 gpurun_out/r4c/bench2.json
