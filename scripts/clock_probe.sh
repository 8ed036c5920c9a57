#!/bin/bash
# sample the shader clock / power while the default bench runs (is the step power- or clock-limited?)
python bench.py --steps 400 --warmup 5 --no-cpu-baseline --profile-every 0 > /tmp/b.json 2>/dev/null &
BP=$!
sleep 12
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|mclk\|power\|Temperature (Sensor junction)\|edge" | tr '\n' ' '; echo
  sleep 1.5
done
wait $BP
grep -o '"ms_per_step": [0-9.]*' /tmp/b.json
