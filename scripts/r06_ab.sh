#!/bin/bash
# Round-6 same-box A/B runs (through gpurun).  Usage: bash scripts/r06_ab.sh <mode> [reps]
#   tune      autotune experiments of round-5 verdict item 1A on the experiment build (scripts/ab/lib_exp.so =
#             TBN_EXPERIMENT=1 python -m attention_based_tbn_amd.build): shipped choices | no <1,1> tiles
#             (TBN_TUNE_MIN_TILE=2) | candidates timed as three concurrent copies (TBN_TUNE_CORUN=2) | both
#   graph2    config 2: shipped stream policy vs whole-step hipGraph replay (verdict item 6)
#   red       the fused BN-backward reduce epilogue on / off per launch
#   sched     the round's two scheduling changes: shipped | weight copies at the start of backward (--no-early-flip) | and the stems'
#             weight gradients in place (--stem-wgrad-last none), config 4 (+ config 3 for the first two arms)
#   turn      length of the forward -> backward turn from the library timeline, with and without the early weight copies
#   bins      the step's timeline per millisecond (conv GEMMs in flight)          -> gpurun_out/r06_timeline_bins_config4.txt
#   batch     config 4 at B = 24 ... 96 clips (own tuned plans each)              -> gpurun_out/r06_batch_sweep.txt
#   wgtraffic fabric reads per weight-gradient launch, per layer shape (counters) -> gpurun_out/r06_wgrad_traffic_per_layer.txt
#   final     closing pass on one box: the whole -m gpu suite, the HBM-traffic counters of the tree (copied into profiles/ ON THE
#             BOX so the lines that follow carry roofline.traffic), three driver-command lines + the default line
# Boxes differ by 2-3 %: only the alternations on ONE box compare.
set -o pipefail
MODE=${1:?mode}; REPS=${2:-3}
EXP=$PWD/scripts/ab/lib_exp.so
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-34s %.3f ms/step  %.2f clips/s  e2e %.4f  conv one-stream %.4f  timed schedule %s' % (sys.argv[1], d['ms_per_step'], d['value'], r['end_to_end_frac'], r['all_conv_gemm']['frac'], (r.get('conv_stage_timed_schedule') or {}).get('frac')))" "$1"; }
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 1"
case $MODE in
  tune)
    for rep in $(seq $REPS); do
      TBN_LIB=$EXP $B 2>/dev/null | line "shipped choices"
      TBN_LIB=$EXP TBN_TUNE_MIN_TILE=2 $B 2>/dev/null | line "no <1,1> tiles"
      TBN_LIB=$EXP TBN_TUNE_CORUN=2 $B 2>/dev/null | line "tuned as 3 concurrent copies"
      TBN_LIB=$EXP TBN_TUNE_CORUN=1 $B 2>/dev/null | line "tuned as 2 concurrent copies"
    done ;;
  graph2)
    ms() { grep -o '"ms_per_step": [0-9.]*'; }
    for rep in $(seq $REPS); do
      echo "config 2 shipped policy (branch + weight-gradient streams), eager   $(python bench.py --config 2 --steps 60 --warmup 5 --no-cpu-baseline --profile-steps 0 --timeline-steps 0 2>/dev/null | ms)"
      echo "config 2 one chain + riders, one stream: $(CFG=2 MULTI=0 AUX=0 BRANCH=0 python scripts/graph_experiment.py 2>/dev/null | grep -E 'eager|graph replay' | tr '\n' ' ')"
    done ;;
  red)
    python scripts/red_epilogue_cost.py 96 ;;
  sched)
    Q="--steps 40 --warmup 5 --no-cpu-baseline --timeline-steps 0 --profile-steps 0"
    ms() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-46s %8.2f clips/s %7.3f ms/step' % (sys.argv[1], d['value'], d['ms_per_step']))" "$1"; }
    for rep in $(seq $REPS); do
      python bench.py $Q 2>/dev/null | ms "shipped"
      python bench.py $Q --no-early-flip 2>/dev/null | ms "--no-early-flip"
      python bench.py $Q --no-early-flip --stem-wgrad-last none 2>/dev/null | ms "--no-early-flip --stem-wgrad-last none"
    done
    for rep in $(seq $REPS); do
      python bench.py --config 3 $Q 2>/dev/null | ms "config 3 shipped"
      python bench.py --config 3 $Q --no-early-flip 2>/dev/null | ms "config 3 --no-early-flip"
    done ;;
  turn)
    for arm in early late; do
      flag=""; [ $arm = late ] && flag="--no-early-flip"
      python bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 --timeline-steps 0 --timeline /tmp/tl_$arm.csv $flag > /dev/null 2>&1
      echo "## $arm"; python scripts/step_timeline.py /tmp/tl_$arm.csv 1 turn | sed -n '/^turn/,$p'
    done ;;
  bins)
    python bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 --timeline-steps 0 --timeline /tmp/tl_4.csv > /dev/null 2>&1
    python scripts/step_timeline.py /tmp/tl_4.csv 1 bins | tee gpurun_out/r06_timeline_bins_config4.txt ;;
  batch)
    for b in 32 24 40 48 64 96 32; do
      python bench.py --batch-per-gpu $b --steps 20 --warmup 5 --no-cpu-baseline --timeline-steps 2 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read()); r = d['roofline']
print('B=%3s  %8.2f clips/s  %7.3f ms/step  end to end %.4f  conv stage timed %.4f  one stream %.4f  dominant %.4f' % (sys.argv[1], d['value'], d['ms_per_step'], r['end_to_end_frac'], r['frac'], r['all_conv_gemm']['frac'], r['dominant']['frac']))" $b | tee -a gpurun_out/r06_batch_sweep.txt
    done ;;
  wgtraffic)
    ROOT=$PWD; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmc_wg
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum -d /tmp/pmc_wg -o t --output-format csv -- python3 $ROOT/scripts/wgrad_traffic.py > /tmp/wg.log 2>&1 || { tail -20 /tmp/wg.log; exit 1; }
    python3 $ROOT/scripts/wgrad_traffic.py "$(find /tmp/pmc_wg -name '*counter_collection.csv' | head -1)" | tee $ROOT/gpurun_out/r06_wgrad_traffic_per_layer.txt ;;
  final)
    set -e
    bash scripts/gpu_full_suite.sh r06_final
    grep -q "pytest rc=0" gpurun_out/r06_final_pytest_full.log
    bash scripts/refresh_profiles.sh r06 traffic > gpurun_out/r06_final_traffic.log 2>&1
    cp gpurun_out/r06_pmc_traffic.json gpurun_out/r06_pmc_traffic_top.txt profiles/
    for i in 1 2 3; do
      timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_final_bench_driver_command_$i.json 2> /dev/null
      head -c 160 gpurun_out/r06_final_bench_driver_command_$i.json; echo
    done
    timeout -k 10 400 python bench.py > gpurun_out/r06_final_bench_default.json 2> /dev/null ;;
  *) echo "unknown mode $MODE"; exit 2 ;;
esac
