#!/bin/bash
# VERDICT r5 #2, the what-if: would a third wave per SIMD (ten compute waves) pay if a smaller operand ring left the LDS for it?
# Build first (container):  bash tools/build_variant.sh w20 -DSFA_WHATIF_SHAPE20 -DSFA_X_OPR_CUT=18
# On the GPU box: the shipped seven-stage shape (14, nine waves) from the product library against shape 20 (2,2,2,2,2,1,1,1,1,1 on twelve waves, three per SIMD; its ring cut to
# what the LDS holds: wrong values, the real instruction stream) from the what-if library, same box, the solver alone at 16 / 64 / 128 windows.
X=slowflow_amd/csrc/build_x
echo "== product library, shape 14 (nine waves: 2,2,2,2,2,2,3)"
SFA_DEBUG=1 timeout -k 10 300 python3 tools/bench_sor_chain.py "16 64 128" "14" || exit 1
echo "== what-if library (ring cut by 18 rows), shape 20 (twelve waves: 2,2,2,2,2,1,1,1,1,1) and, for the cut ring's own effect, shape 14"
SFA_DEBUG=1 SFA_LIB=$X/libsfa_w20.so timeout -k 10 300 python3 tools/bench_sor_chain.py "16 64 128" "20 14" || exit 1
echo "== product library again"
SFA_DEBUG=1 timeout -k 10 300 python3 tools/bench_sor_chain.py "16 64 128" "14" || exit 1
