set -e
mkdir -p gpurun_out/r04_summary
( timeout -k 10 200 python3 tools/soak.py 150 64; timeout -k 10 200 python3 tools/soak.py 20 16 cfg ) > gpurun_out/r04_summary/r04_soak.txt 2>&1
timeout -k 10 400 python3 tools/bench_configs.py > gpurun_out/r04_summary/r04_other_configs.txt 2>&1
bash profiles/collect_config5.sh r04 32 > gpurun_out/r04_c5.log 2>&1
timeout -k 10 300 python3 tools/run_driver_cfg4.py 64 > gpurun_out/r04_summary/r04_cfg4_driver.txt 2>&1
echo done2
