#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_model_gpu.py tests/test_trainstep_gpu.py -q -m gpu --maxfail=10 -s > gpurun_out/c4_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a gpurun_out/c4_pytest.log
grep -n "R = 96\|oracle step\|kernel families\|forced-decision\|passed\|failed" gpurun_out/c4_pytest.log | tail -20
if [ $rc -ge 124 ]; then exit $rc; fi
for rows in 0 1; do
  echo "rows=$rows $(TBN_STEM_ROWS=$rows timeout -k 10 120 python scripts/layer_profile.py 10 224 224 96 2>/dev/null | tee gpurun_out/c4_lp_rows${rows}.txt | grep 'total conv')"
  grep conv1_7x7 gpurun_out/c4_lp_rows${rows}.txt
done
for rep in 1 2 3; do
  for rows in 0 1; do
    echo "rows=$rows $(TBN_STEM_ROWS=$rows python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-every 0 2>/dev/null | grep -o 'ms_per_step.: [0-9.]*')"
  done
done
