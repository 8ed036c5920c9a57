#!/bin/bash
# Regenerates the per-round measurement artifacts on the GPU box into gpurun_out/<tag>_* (copy what is to be judged
# into profiles/).  Usage (through gpurun): bash scripts/refresh_profiles.sh r01_final9
set -eo pipefail
TAG=${1:?tag}
O=gpurun_out
mkdir -p $O
timeout -k 10 400 python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
for c in 2 3 5; do
  timeout -k 10 300 python bench.py --config $c > $O/${TAG}_bench_config$c.json 2> /dev/null
done
timeout -k 10 300 python bench.py --forward-only > $O/${TAG}_bench_config4_forward_only.json 2> /dev/null
timeout -k 10 120 python scripts/layer_profile.py 3 224 224 96 > $O/${TAG}_layer_profile_rgb_R96_single_stream.txt 2> /dev/null
timeout -k 10 120 python scripts/layer_profile.py 10 224 224 96 > $O/${TAG}_layer_profile_flow_R96_single_stream.txt 2> /dev/null
timeout -k 10 120 python scripts/layer_profile.py 1 256 256 96 > $O/${TAG}_layer_profile_audio_R96_single_stream.txt 2> /dev/null
# rocprofv3 kernel trace of the bench command with HIP-event brackets on every step (same kernels, same averages)
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o trace --output-format csv -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-every 1 > $ROOT/$O/${TAG}_bench_profile_every_1.json 2> $ROOT/$O/${TAG}_rocprof.err
cp "$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1)" $ROOT/$O/${TAG}_bench_profile_every_1.kernel_stats.csv
cd $ROOT
head -c 250 $O/${TAG}_bench_default.json; echo
