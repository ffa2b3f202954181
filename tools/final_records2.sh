# the other records of a round (soak, the other BASELINE configurations, config 5's HBM report, config 4 through the driver, the cut's share): bash tools/final_records2.sh TAG HEAD
set -e
TAG=${1:-r06}; HEAD=${2:-unknown}
mkdir -p gpurun_out/${TAG}_summary
( echo "# HEAD $HEAD"; timeout -k 10 200 python3 tools/soak.py 150 64; timeout -k 10 300 python3 tools/soak.py 60 128; timeout -k 10 200 python3 tools/soak.py 20 16 cfg ) > gpurun_out/${TAG}_summary/${TAG}_soak.txt 2>&1
( echo "# HEAD $HEAD"; timeout -k 10 400 python3 tools/bench_configs.py ) > gpurun_out/${TAG}_summary/${TAG}_other_configs.txt 2>&1
bash profiles/collect_config5.sh $TAG 32 > gpurun_out/${TAG}_c5.log 2>&1
( echo "# HEAD $HEAD"; timeout -k 10 300 python3 tools/run_driver_cfg4.py 64 ) > gpurun_out/${TAG}_summary/${TAG}_cfg4_driver.txt 2>&1
( echo "# HEAD $HEAD"; bash tools/profile_cut.sh $TAG ) > gpurun_out/${TAG}_summary/${TAG}_cut_profile.txt 2>&1
echo done2
