#!/bin/bash
# Regenerates the per-round measurement artifacts on the GPU box into gpurun_out/<tag>_* (copy what is to be judged
# into profiles/).  Usage (through gpurun): bash scripts/refresh_profiles.sh r02   [quick]
# Sections: bench lines (default / other configs / forward only / STFT inputs), per-layer profiles, rocprofv3 kernel
# stats of the bench command and of config 3 (heads), STFT kernel, HBM traffic (FETCH_SIZE / WRITE_SIZE: separate
# passes) and SQ counters (MFMA utilisation) per modality.
set -eo pipefail
TAG=${1:?tag}
QUICK=${2:-}
O=gpurun_out
ROOT=$PWD
mkdir -p $O
timeout -k 10 400 python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
echo "default: $(head -c 200 $O/${TAG}_bench_default.json)"
if [ -z "$QUICK" ]; then
  for c in 2 3 5; do
    timeout -k 10 300 python bench.py --config $c > $O/${TAG}_bench_config$c.json 2> /dev/null
  done
  timeout -k 10 300 python bench.py --forward-only > $O/${TAG}_bench_config4_forward_only.json 2> /dev/null
  timeout -k 10 300 python bench.py --stft-inputs --no-cpu-baseline > $O/${TAG}_bench_stft_inputs.json 2> /dev/null
  timeout -k 10 300 python bench.py --config 3 --stft-inputs > $O/${TAG}_bench_config3_stft_inputs.json 2> /dev/null
fi
timeout -k 10 120 python scripts/layer_profile.py 3 224 224 96 > $O/${TAG}_layer_profile_rgb_R96_single_stream.txt 2> /dev/null
timeout -k 10 120 python scripts/layer_profile.py 10 224 224 96 > $O/${TAG}_layer_profile_flow_R96_single_stream.txt 2> /dev/null
timeout -k 10 120 python scripts/layer_profile.py 1 256 256 96 > $O/${TAG}_layer_profile_audio_R96_single_stream.txt 2> /dev/null
timeout -k 10 120 python scripts/stft_profile.py > $O/${TAG}_stft_profile.txt 2> /dev/null
cd /tmp && export TMPDIR=/tmp
# rocprofv3 kernel trace of the bench command with HIP-event brackets on every step (same kernels, same averages)
rm -rf /tmp/prof_$TAG
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o trace --output-format csv -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-every 1 > $ROOT/$O/${TAG}_bench_profile_every_1.json 2> $ROOT/$O/${TAG}_rocprof.err
cp "$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1)" $ROOT/$O/${TAG}_bench_profile_every_1.kernel_stats.csv
# heads (PE / GroupNorm / MHA / fusion) + STFT: config 3 from waveforms
rm -rf /tmp/prof_h_$TAG
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_h_$TAG -o trace --output-format csv -- python3 $ROOT/bench.py --config 3 --stft-inputs --steps 10 --warmup 3 --no-cpu-baseline --profile-every 0 > $ROOT/$O/${TAG}_heads_bench_config3_stft.json 2> /dev/null
cp "$(find /tmp/prof_h_$TAG -name '*kernel_stats.csv' | head -1)" $ROOT/$O/${TAG}_heads_config3_stft.kernel_stats.csv
# HBM traffic: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), counters only + kernel trace
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o t --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --profile-every 1 > /dev/null 2>&1
done
python3 $ROOT/scripts/pmc_traffic.py /tmp/pmc_FETCH_SIZE/t_counter_collection.csv /tmp/pmc_WRITE_SIZE/t_counter_collection.csv $ROOT/$O/${TAG}_pmc_traffic.json 3 "$(cd $ROOT && cat .gitrev 2>/dev/null)" > $ROOT/$O/${TAG}_pmc_traffic_top.txt
# heads / STFT traffic (HBM-bound kernels of config 3 from waveforms)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmch_$c
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmch_$c -o t --output-format csv -- python3 $ROOT/bench.py --config 3 --stft-inputs --steps 3 --warmup 2 --no-cpu-baseline --profile-every 0 > /dev/null 2>&1
done
python3 $ROOT/scripts/pmc_traffic.py /tmp/pmch_FETCH_SIZE/t_counter_collection.csv /tmp/pmch_WRITE_SIZE/t_counter_collection.csv $ROOT/$O/${TAG}_heads_pmc_traffic.json 3 > /dev/null
# SQ counters per modality (MFMA pipe utilisation per kernel, wave-level wait breakdown)
for m in "rgb 3 224 224" "flow 10 224 224" "audio 1 256 256"; do
  set -- $m
  rm -rf /tmp/pmcsq_$1
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU -d /tmp/pmcsq_$1 -o t --output-format csv -- python3 $ROOT/scripts/layer_profile.py $2 $3 $4 96 > /dev/null 2>&1
  python3 $ROOT/scripts/pmc_sq.py /tmp/pmcsq_$1/t_counter_collection.csv 40 > $ROOT/$O/${TAG}_pmc_sq_$1.txt
done
cd $ROOT
echo done
