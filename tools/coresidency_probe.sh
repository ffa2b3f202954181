#!/bin/bash
# VERDICT r4 #1 what-if: can the solver and the data-term kernel share a CU?  Whole-path ms per step for solver shapes whose workgroup leaves room for one
# 48-KB block of k_assemble_images (LDS granule 1280 B: solver <= 90 granules), two lockstep groups on two streams (overlap possible) and one group on one stream
# (no overlap possible: what the shape itself costs).  usage (GPU box): bash tools/coresidency_probe.sh > gpurun_out/coresidency.txt
X=slowflow_amd/csrc/build_x
run() {  # label, batch, streams
  timeout -k 10 300 python3 bench.py --batch $2 --streams $3 --steps 5 --warmup 2 --no-cpu-baseline --path-only | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2 x $3:', round(d['value']), d['ms_per_step'], flush=True)" || exit 1; }
both() { run "$1" 128 2 && run "$1" 64 1; }
export SFA_DEBUG=1   # the library reads its switches from the environment only behind this
both default || exit 1
( export SFA_SOR_CHAIN=5; both "shape5(2x5,KG10,117KB:no-room)" ) || exit 1
( export SFA_SOR_CHAIN=5 SFA_LIB=$X/libsfa_oprcut3.so; both "shape5-oprcut3(110KB:room-for-1-asm-block;what-if)" ) || exit 1
( export SFA_SOR_CHAIN=6; both "shape6(1x5,KG5,89KB:room)" ) || exit 1
( export SFA_SOR_CHAIN=2; both "shape2(2x3,KG6,73KB:room)" ) || exit 1
( export SFA_SOR_CHAIN=8; both "shape8(3x2,KG6,63KB:room)" ) || exit 1
both default
