#!/bin/bash
# same-box A/B of the lone solve: tools/ab_lone.sh NAME ...  (build_x/libsfa_NAME.so; "default" = the product)
X=slowflow_amd/csrc/build_x
for n in "$@"; do
  if [ $n = default ]; then unset SFA_LIB; else export SFA_LIB=$X/libsfa_$n.so; fi
  echo "== $n"; timeout -k 10 300 python3 tools/bench_sor_chain.py "1 4 8" "16 6" | grep -v amdgpu.ids || exit 1
done
