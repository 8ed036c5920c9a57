#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O
for i in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg4 20 steps', d['value'], d['ms_per_step'], d['step_gpu_ms'])"
done
python bench.py --config 3 --steps 100 --warmup 10 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 100 steps', d['value'], d['ms_per_step'], d['step_gpu_ms'])"
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg4 100 steps', d['value'], d['ms_per_step'], d['step_gpu_ms'])"
echo done
