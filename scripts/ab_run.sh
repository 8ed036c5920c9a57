#!/bin/bash
# same-box A/B of two builds of libtbn_hip.so (put them at scripts/ab/lib_A.so, scripts/ab/lib_B.so -- git-ignored, but
# they travel with gpurun): alternates them under the default multi-stream bench.  Boxes differ by +-2 %, so only
# same-box comparisons resolve changes of ~1 %.
L=attention_based_tbn_amd/libtbn_hip.so
cp $L /tmp/orig.so
for rep in 1 2 3; do
  for v in A B; do
    cp scripts/ab/lib_$v.so $L
    echo "$v $(python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 0 2>/dev/null | grep -o 'ms_per_step.: [0-9.]*')"
  done
done
cp /tmp/orig.so $L
