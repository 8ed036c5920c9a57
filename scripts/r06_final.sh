#!/bin/bash
# round-6 closing pass on one box: the whole -m gpu suite, then the HBM-traffic counters of the final tree, then the headline lines
set -eo pipefail
bash scripts/gpu_full_suite.sh r06n
grep -q "pytest rc=0" gpurun_out/r06n_pytest_full.log
bash scripts/refresh_profiles.sh r06 traffic > gpurun_out/r06n_traffic.log 2>&1
cp gpurun_out/r06_pmc_traffic.json profiles/r06_pmc_traffic.json
for i in 1 2 3; do
  timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06n_bench_driver_command_$i.json 2> /dev/null
  head -c 160 gpurun_out/r06n_bench_driver_command_$i.json; echo
done
timeout -k 10 400 python bench.py > gpurun_out/r06n_bench_default.json 2> /dev/null
for c in 2 3; do timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > gpurun_out/r06n_bench_config$c.json 2> /dev/null; done
timeout -k 10 300 python bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 --timeline-steps 0 --timeline /tmp/tl_4.csv > /dev/null 2>&1
python scripts/step_timeline.py /tmp/tl_4.csv 1 > gpurun_out/r06n_timeline_config4.txt 2>&1
echo final-done
