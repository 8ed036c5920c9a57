#!/bin/bash
set -eo pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -q -x -m gpu -k "early_weight_flip or stem_weight or golden or capture" > gpurun_out/r06q_pytest.log 2>&1 || { tail -30 gpurun_out/r06q_pytest.log; exit 1; }
tail -3 gpurun_out/r06q_pytest.log
for i in 1 2 3 4 5; do
  for arm in "" "--no-early-flip"; do
    timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --timeline-steps 0 --profile-steps 0 $arm > /tmp/b.json 2> /dev/null
    python - "$arm" <<'PY' | tee -a gpurun_out/r06q_ab_early_flip.txt
import json, sys
d = json.load(open('/tmp/b.json')); print("%-16s %8.2f clips/s %7.3f ms/step" % (sys.argv[1] or "early (shipped)", d['value'], d['ms_per_step']))
PY
  done
done
for i in 1 2 3; do
  for arm in "" "--no-early-flip"; do
    timeout -k 10 300 python bench.py --config 3 --steps 30 --warmup 5 --no-cpu-baseline --timeline-steps 0 --profile-steps 0 $arm > /tmp/b.json 2> /dev/null
    python - "config 3 $arm" <<'PY' | tee -a gpurun_out/r06q_ab_early_flip.txt
import json, sys
d = json.load(open('/tmp/b.json')); print("%-26s %8.2f clips/s %7.3f ms/step" % (sys.argv[1], d['value'], d['ms_per_step']))
PY
  done
done
