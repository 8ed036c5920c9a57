#!/bin/bash
mkdir -p gpurun_out
ROOT=$PWD
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/c12_bench_driver_cmd.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/c12_bench_driver_cmd.json'))
r=d['roofline']
print('driver cmd:', d['value'], d['ms_per_step'], 'dominant', r['kernel'], r['frac'], r['avg_launch_us'], 'all', r['all_conv_gemm'], 'e2e', r['end_to_end_frac'])
PY
for m in "3 224 224" "10 224 224" "1 256 256"; do
  echo "cin=$(echo $m | cut -d' ' -f1) $(timeout -k 10 120 python scripts/layer_profile.py $m 96 2>/dev/null | grep 'total conv')"
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof12
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/prof12 -o trace --output-format csv -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-every 1 > $ROOT/gpurun_out/c12_bench_profile_every_1.json 2> /dev/null
cp "$(find /tmp/prof12 -name '*kernel_stats.csv' | head -1)" $ROOT/gpurun_out/c12_kernel_stats.csv
cd $ROOT
python - <<'PY'
import json,csv
d=json.load(open('gpurun_out/c12_bench_profile_every_1.json'))
r=d['roofline']
print('under rocprof, events:', r['kernel'], r['avg_launch_us'], 'us')
for row in csv.DictReader(open('gpurun_out/c12_kernel_stats.csv')):
    if 'conv_wgrad_kernel<2, 2, 0>' in row['Name'] or 'conv_wgrad_kernel<2, 2, 2>' in row['Name']:
        print('rocprof:', row['Name'][:40], row['Calls'], float(row['AverageNs'])/1e3, 'us')
for e in r['by_kernel'][:4]: print('   events', e)
PY
