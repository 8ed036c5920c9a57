#!/bin/bash
# same-box A/B of an environment knob under the default multi-stream bench: bash scripts/ab_env.sh VAR v1 v2 [v3 ...]
# (three alternating rounds; boxes differ by +-2-3 %, so only same-box comparisons resolve changes of ~1 %)
VAR=${1:?variable}; shift
for rep in 1 2 3; do
  for v in "$@"; do
    echo "$VAR=$v $(env $VAR=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-every 0 2>/dev/null | grep -o 'ms_per_step.: [0-9.]*')"
  done
done
