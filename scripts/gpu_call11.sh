#!/bin/bash
for rep in 1 2 3; do
  for b in 0 5 12; do
    echo "s1_bias=$b $(TBN_TUNE_S1_BIAS=$b python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-every 0 2>/dev/null | grep -o 'ms_per_step.: [0-9.]*')"
  done
done
