#!/bin/bash
# The "HBM-roofline report" BASELINE configs[4] asks for: per kernel of one config-5 refinement (2048x2048, 6 levels, Lorentzian, batch 32 on one stream) the
# launch duration (kernel trace), the HBM-side bytes (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes as MI355X_MICROARCH.md prescribes) and the VALU share.
#   bash profiles/collect_config5.sh r04 [batch]   -> profiles/<tag>_config5_hbm_report.txt   (run on the GPU box from the repo root)
set -e
TAG=${1:-r04}
B=${2:-32}
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_c5_stats -o c5 -f csv -- python3 $R/tools/bench_config5_step.py $B > $R/gpurun_out/${TAG}_c5_stats.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_c5_fetch -o c5 -f csv -- python3 $R/tools/bench_config5_step.py $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_c5_write -o c5 -f csv -- python3 $R/tools/bench_config5_step.py $B > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_c5_sq -o c5 -f csv -- python3 $R/tools/bench_config5_step.py $B > /dev/null 2>&1
cd $R
python3 tools/bench_config5_step.py $B > gpurun_out/${TAG}_c5_plain.txt 2>&1
python3 tools/config5_report.py $TAG $B > profiles/${TAG}_config5_hbm_report.txt
rm -rf gpurun_out/${TAG}_c5_stats gpurun_out/${TAG}_c5_fetch gpurun_out/${TAG}_c5_write gpurun_out/${TAG}_c5_sq
mkdir -p gpurun_out/${TAG}_summary && cp profiles/${TAG}_config5_hbm_report.txt gpurun_out/${TAG}_summary/
cat profiles/${TAG}_config5_hbm_report.txt
