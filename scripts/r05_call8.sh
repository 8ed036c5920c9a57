#!/bin/bash
set -o pipefail
python scripts/debug/straggler_probe.py 3 2>&1 | grep -v "Freezing\|amdgpu.ids" | head -12
timeout -k 10 300 python -m pytest tests/test_model_gpu.py -q -m gpu -k "golden or train_step_matches" 2>&1 | tail -3
for i in 1 2; do python bench.py --config 3 --steps 100 --warmup 10 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg3', d['value'], d['ms_per_step'], d['step_gpu_ms'])"; done
python bench.py --config 3 --audio-2p1s --steps 60 --warmup 10 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cfg3 2.1s', d['value'], d['ms_per_step'], d['step_gpu_ms']['median'])"
