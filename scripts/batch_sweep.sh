#!/bin/bash
# config 4 at other batch sizes (own autotuned plan each): is the step limited by grid quantisation at 96 frames per modality?
mkdir -p gpurun_out
for b in 32 24 40 48 64 96 32; do
  timeout -k 10 300 python bench.py --batch-per-gpu $b --steps 20 --warmup 5 --no-cpu-baseline --timeline-steps 2 > /tmp/bs.json 2> /dev/null || { echo "B=$b failed"; continue; }
  python - $b <<'PY' | tee -a gpurun_out/r06o_batch_sweep.txt
import json, sys
d = json.load(open('/tmp/bs.json')); r = d['roofline']
print("B=%3s  %8.2f clips/s  %7.3f ms/step  end to end %.4f  conv stage timed %.4f  one stream %.4f  dominant %.4f" % (sys.argv[1], d['value'], d['ms_per_step'], r['end_to_end_frac'], r['frac'], r['all_conv_gemm']['frac'], r['dominant']['frac']))
PY
done
