#!/bin/bash
# same-box A/B of an environment knob under the default multi-stream bench: bash scripts/ab_env.sh VAR v1 v2 [v3 ...]
# (round 6: VAR is only read by the experiment build -- TBN_EXPERIMENT=1 python -m attention_based_tbn_amd.build -- so the runs load scripts/ab/lib_exp.so)
# (three alternating rounds; boxes differ by +-2-3 %, so only same-box comparisons resolve changes of ~1 %)
VAR=${1:?variable}; shift
for rep in 1 2 3; do
  for v in "$@"; do
    echo "$VAR=$v $(env TBN_LIB=$PWD/scripts/ab/lib_exp.so $VAR=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-every 0 2>/dev/null | grep -o 'ms_per_step.: [0-9.]*')"
  done
done
