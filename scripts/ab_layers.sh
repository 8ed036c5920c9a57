#!/bin/bash
# same-box A/B of two library builds (scripts/ab/lib_A.so, lib_B.so, selected with TBN_LIB): one-stream conv-GEMM totals of
# the three backbones (scripts/layer_profile.py), then the default multi-stream bench alternated 3x (scripts/ab_run.sh)
set -o pipefail
mkdir -p gpurun_out
for v in A B; do
  for m in "3 224 224" "10 224 224" "1 256 256"; do
    TBN_LIB=$PWD/scripts/ab/lib_$v.so timeout -k 10 120 python scripts/layer_profile.py $m 96 2>/dev/null > gpurun_out/lp_${v}_$(echo $m | cut -d" " -f1).txt || exit 1
    echo "$v cin=$(echo $m | cut -d' ' -f1) $(grep 'total conv' gpurun_out/lp_${v}_$(echo $m | cut -d' ' -f1).txt)"
  done
done
bash scripts/ab_run.sh
