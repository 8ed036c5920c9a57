#!/bin/bash
# Round-6 same-box A/B runs (through gpurun).  Usage: bash scripts/r06_ab.sh <mode> [reps]
#   tune      autotune experiments of round-5 verdict item 1A on the experiment build (scripts/ab/lib_exp.so =
#             TBN_EXPERIMENT=1 python -m attention_based_tbn_amd.build): shipped choices | no <1,1> tiles
#             (TBN_TUNE_MIN_TILE=2) | candidates timed as three concurrent copies (TBN_TUNE_CORUN=2) | both
#   graph2    config 2: shipped stream policy vs whole-step hipGraph replay (verdict item 6)
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
esac
