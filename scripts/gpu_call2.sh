#!/bin/bash
# round-3 call 2: full GPU parity suite of the new build, then the wgrad split-K plan with 4 workgroups per CU assumed
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu --maxfail=8 > gpurun_out/c2_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a gpurun_out/c2_pytest.log
tail -12 gpurun_out/c2_pytest.log
if [ $rc -ge 124 ]; then exit $rc; fi
for occ in 0 3 4; do
  for m in "3 224 224" "1 256 256"; do
    echo "occ=$occ cin=$(echo $m | cut -d' ' -f1) $(TBN_WGRAD_OCC=$occ timeout -k 10 120 python scripts/layer_profile.py $m 96 2>/dev/null | tee gpurun_out/c2_lp_occ${occ}_$(echo $m | cut -d' ' -f1).txt | grep 'total conv')"
  done
done
for rep in 1 2; do
  for occ in 0 4; do
    echo "occ=$occ $(TBN_WGRAD_OCC=$occ python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-every 0 2>/dev/null | grep -o 'ms_per_step.: [0-9.]*')"
  done
done
