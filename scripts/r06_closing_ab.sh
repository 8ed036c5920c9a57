#!/bin/bash
# same-box closing A/B of the round's two scheduling changes: shipped | copies at the start of backward | also the stems' weight gradients in place
set -eo pipefail
mkdir -p gpurun_out
for i in 1 2 3 4; do
  for arm in "shipped" "--no-early-flip" "--no-early-flip --stem-wgrad-last none"; do
    flags="$arm"; [ "$arm" = shipped ] && flags=""
    timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --timeline-steps 0 --profile-steps 0 $flags > /tmp/b.json 2> /dev/null
    python - "$arm" <<'PY' | tee -a gpurun_out/r06s_closing_ab.txt
import json, sys
d = json.load(open('/tmp/b.json')); print("%-42s %8.2f clips/s %7.3f ms/step" % (sys.argv[1], d['value'], d['ms_per_step']))
PY
  done
done
for i in 1 2 3; do
  timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06s_bench_driver_command_$i.json 2> /dev/null
  head -c 160 gpurun_out/r06s_bench_driver_command_$i.json; echo
done
