#!/bin/bash
# same-box A/B of two builds of libtbn_hip.so (scripts/ab/lib_A.so, scripts/ab/lib_B.so -- git-ignored, but they travel with
# gpurun; build them with TBN_BUILD_VARIANT=A / =B python -m attention_based_tbn_amd.build): alternates them under the
# default multi-stream bench, selected with TBN_LIB -- the shipped library is never touched.  Boxes differ by +-2 %, so
# only same-box comparisons resolve changes of ~1 %.  Usage: ab_run.sh [extra bench.py arguments]
set -o pipefail
for rep in 1 2 3; do
  for v in A B; do
    echo "$v $(TBN_LIB=$PWD/scripts/ab/lib_$v.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 0 "$@" 2>/dev/null | grep -o 'ms_per_step.: [0-9.]*')"
  done
done
