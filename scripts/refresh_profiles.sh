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
if [ "$QUICK" = "traffic" ]; then ONLY_TRAFFIC=1; fi
if [ "$QUICK" = "pmc" ]; then ONLY_PMC=1; QUICK=; fi
if [ "$QUICK" = "bench" ]; then ONLY_BENCH=1; QUICK=; fi
if [ -z "$ONLY_TRAFFIC" ] && [ -z "$ONLY_PMC" ]; then
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
echo "default: $(head -c 200 $O/${TAG}_bench_default.json)"
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_driver_command.json 2> /dev/null
echo "driver command: $(head -c 200 $O/${TAG}_bench_driver_command.json)"
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
# the un-traced multi-stream step on the library's own kernel timeline: critical-path attribution (config 4 and the lone backbone)
for c in 4 2; do
  timeout -k 10 300 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 --timeline-steps 0 --timeline /tmp/tl_$c.csv > /dev/null 2>&1
  python scripts/step_timeline.py /tmp/tl_$c.csv 1 > $O/${TAG}_timeline_config$c.txt 2>&1
done
timeout -k 10 200 python scripts/red_epilogue_cost.py 96 2> /dev/null | grep -v amdgpu.ids > $O/${TAG}_red_epilogue_cost.txt
cd /tmp && export TMPDIR=/tmp
# rocprofv3 kernel trace of the bench command with HIP-event brackets on every step (same kernels, same averages)
rm -rf /tmp/prof_$TAG
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o trace --output-format csv -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-every 1 --timeline-steps 0 > $ROOT/$O/${TAG}_bench_profile_every_1.json 2> $ROOT/$O/${TAG}_rocprof.err
cp "$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1)" $ROOT/$O/${TAG}_bench_profile_every_1.kernel_stats.csv
# heads (PE / GroupNorm / MHA / fusion) + STFT: config 3 from waveforms
rm -rf /tmp/prof_h_$TAG
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_h_$TAG -o trace --output-format csv -- python3 $ROOT/bench.py --config 3 --stft-inputs --steps 10 --warmup 3 --no-cpu-baseline --profile-every 0 --timeline-steps 0 > $ROOT/$O/${TAG}_heads_bench_config3_stft.json 2> /dev/null
cp "$(find /tmp/prof_h_$TAG -name '*kernel_stats.csv' | head -1)" $ROOT/$O/${TAG}_heads_config3_stft.kernel_stats.csv
fi
if [ -n "$ONLY_BENCH" ]; then cd $ROOT; echo done; exit 0; fi
cd /tmp && export TMPDIR=/tmp
# HBM traffic: L2 -> fabric read requests by size class (one pass: three TCC counters) and WRITE_SIZE (its own pass), priced
# with bytes per request calibrated on launches of known byte counts (scripts/pmc_calibrate.py, same three counters)
traffic() {
  rm -rf /tmp/pmc_cal /tmp/pmc_RD /tmp/pmc_WRITE_SIZE
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum -d /tmp/pmc_cal -o t --output-format csv -- python3 $ROOT/scripts/pmc_calibrate.py > /tmp/pmc_cal_expect.txt 2> /tmp/pmc_cal.err
  tail -1 /tmp/pmc_cal_expect.txt > /tmp/pmc_cal_expect.json
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum -d /tmp/pmc_RD -o t --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --profile-every 1 --timeline-steps 0 > /dev/null 2>&1
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_WRITE_SIZE -o t --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --profile-every 1 --timeline-steps 0 > /dev/null 2>&1
  python3 $ROOT/scripts/pmc_traffic.py --raw "$(find /tmp/pmc_RD -name '*counter_collection.csv' | head -1)" "$(find /tmp/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)" $ROOT/$O/${TAG}_pmc_traffic.json 3 "$(cd $ROOT && cat .gitrev 2>/dev/null || true)" "$(find /tmp/pmc_cal -name '*counter_collection.csv' | head -1)" /tmp/pmc_cal_expect.json > $ROOT/$O/${TAG}_pmc_traffic_top.txt
}
traffic
if [ -n "$ONLY_TRAFFIC" ]; then cd $ROOT; cat $O/${TAG}_pmc_traffic_top.txt; echo done; exit 0; fi
# heads / STFT traffic (HBM-bound kernels of config 3 from waveforms)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmch_$c
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmch_$c -o t --output-format csv -- python3 $ROOT/bench.py --config 3 --stft-inputs --steps 3 --warmup 2 --no-cpu-baseline --profile-every 0 --timeline-steps 0 > /dev/null 2>&1
done
python3 $ROOT/scripts/pmc_traffic.py /tmp/pmch_FETCH_SIZE/t_counter_collection.csv /tmp/pmch_WRITE_SIZE/t_counter_collection.csv $ROOT/$O/${TAG}_heads_pmc_traffic.json 3 > /dev/null
# SQ counters per modality (MFMA pipe utilisation per kernel, wave-level wait breakdown)
for m in "rgb 3 224 224" "flow 10 224 224" "audio 1 256 256"; do
  set -- $m
  rm -rf /tmp/pmcsq_$1
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU -d /tmp/pmcsq_$1 -o t --output-format csv -- python3 $ROOT/scripts/layer_profile.py $2 $3 $4 96 burst > /dev/null 2>&1
  python3 $ROOT/scripts/pmc_sq.py /tmp/pmcsq_$1/t_counter_collection.csv 40 > $ROOT/$O/${TAG}_pmc_sq_$1.txt
done
cd $ROOT
echo done
