#!/bin/bash
# SQ-side PMC counters of the SOR band kernel alone (tools/bench_sor_batch.py 32) and of the bench's kernels, on the GPU box:
#   bash profiles/collect_sq.sh r02   -> gpurun_out/<tag>_sq{1,2,3}/...   (each --pmc pass in its own run, --kernel-trace only)
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d gpurun_out/${TAG}_sq1 -o sor -f csv -- python3 tools/bench_sor_batch.py 32 > gpurun_out/${TAG}_sq1.txt 2> gpurun_out/${TAG}_sq1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_BRANCH -d gpurun_out/${TAG}_sq2 -o sor -f csv -- python3 tools/bench_sor_batch.py 32 > gpurun_out/${TAG}_sq2.txt 2> gpurun_out/${TAG}_sq2.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE -d gpurun_out/${TAG}_sq3 -o sor -f csv -- python3 tools/bench_sor_batch.py 32 > gpurun_out/${TAG}_sq3.txt 2> gpurun_out/${TAG}_sq3.err
echo done
