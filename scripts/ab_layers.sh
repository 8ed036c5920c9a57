#!/bin/bash
# same-box A/B of two library builds (scripts/ab/lib_A.so, lib_B.so): one-stream conv-GEMM totals of the three backbones
# (scripts/layer_profile.py), then the default multi-stream bench alternated 3x (scripts/ab_run.sh)
L=attention_based_tbn_amd/libtbn_hip.so
cp $L /tmp/orig_layers.so
for v in A B; do
  cp scripts/ab/lib_$v.so $L
  for m in "3 224 224" "10 224 224" "1 256 256"; do
    timeout -k 10 120 python scripts/layer_profile.py $m 96 2>/dev/null > gpurun_out/lp_${v}_$(echo $m | cut -d" " -f1).txt || exit 1
    echo "$v cin=$(echo $m | cut -d' ' -f1) $(grep 'total conv' gpurun_out/lp_${v}_$(echo $m | cut -d' ' -f1).txt)"
  done
done
cp /tmp/orig_layers.so $L
bash scripts/ab_run.sh
