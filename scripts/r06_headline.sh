#!/bin/bash
# three driver-command lines on the final tree (profiles/r06_pmc_traffic.json must carry this tree's source hash: roofline.traffic is then live)
set -eo pipefail
mkdir -p gpurun_out
for i in 1 2 3; do
  timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06t_bench_driver_command_$i.json 2> /dev/null
  python - $i <<'PY'
import json, sys
d = json.load(open('gpurun_out/r06t_bench_driver_command_%s.json' % sys.argv[1])); r = d['roofline']
print(d['value'], d['ms_per_step'], r['end_to_end_frac'], r['frac'], r['all_conv_gemm']['frac'], r['dominant']['frac'], r['traffic'], r.get('traffic_ratio'))
PY
done
timeout -k 10 400 python bench.py > gpurun_out/r06t_bench_default.json 2> /dev/null
