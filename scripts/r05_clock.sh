#!/bin/bash
# shader clock / power under (a) the three-stream step, (b) the one-stream step, (c) config 5 eval
probe() {
  "$@" > /tmp/b.json 2>/dev/null &
  BP=$!
  sleep 14
  for i in 1 2 3 4 5; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Socket Graphics Package Power" | sed 's/.*://' | tr '\n' ' '; echo
    sleep 1.2
  done
  wait $BP
  grep -o '"ms_per_step": [0-9.]*' /tmp/b.json
}
echo "== three modality streams"; probe python bench.py --steps 600 --warmup 5 --no-cpu-baseline --profile-steps 0
echo "== one stream";             probe python bench.py --steps 450 --warmup 5 --no-cpu-baseline --profile-steps 0 --no-multi-stream
echo "== config 5 eval";          probe python bench.py --config 5 --steps 110 --warmup 3 --no-cpu-baseline --profile-steps 0
