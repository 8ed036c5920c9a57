#!/bin/bash
mkdir -p gpurun_out
for rep in 1 2 3; do
  for b in 0 6 15; do
    echo "halo_bias=$b $(TBN_TUNE_HALO_BIAS=$b python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-every 0 2>/dev/null | grep -o 'ms_per_step.: [0-9.]*')"
  done
done
for b in 0 15; do
  echo "halo_bias=$b cin=3 $(TBN_TUNE_HALO_BIAS=$b timeout -k 10 120 python scripts/layer_profile.py 3 224 224 96 2>/dev/null | tee gpurun_out/c9_lp_bias${b}.txt | grep 'total conv')"
done
