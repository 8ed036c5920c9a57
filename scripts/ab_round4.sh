#!/bin/bash
# The same-box A/B runs of round 4 (results under profiles/r04_ab_*.txt).  usage (through gpurun): bash scripts/ab_round4.sh <name>
# (round 6: the environment knobs -- TBN_LPT, TBN_WGRAD_LDS_PAD -- are only read by the experiment build: prefix TBN_LIB=$PWD/scripts/ab/lib_exp.so)
#   lpt       parity-phase / pair launches in parity vs longest-K-first order (TBN_LPT), one-stream backbone totals
#   prio      which modality streams get the high HIP priority (bench.py --high-prio)
#   wpad      weight-gradient kernel capped at two workgroups per CU (TBN_WGRAD_LDS_PAD): step time
#   branch4   branch-level side stream with three modalities (bench.py --branch-streams)
#   chunk     frames per eval engine call of config 5 (bench.py --eval-chunk)
step() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 0 "$@" 2>/dev/null | grep -o 'ms_per_step.: [0-9.]*'; }
case "$1" in
  lpt) for i in 1 2; do for v in 0 1; do
         echo "RGB TBN_LPT=$v $(TBN_LPT=$v python scripts/layer_profile.py 3 224 224 96 2>/dev/null | head -1)"
         echo "Audio TBN_LPT=$v $(TBN_LPT=$v python scripts/layer_profile.py 1 256 256 96 2>/dev/null | head -1)"; done; done ;;
  prio) for i in 1 2 3; do for v in Audio Audio,Flow Flow none; do echo "--high-prio $v $(step --high-prio $v)"; done; done ;;
  wpad) for i in 1 2; do for v in 0 24576; do echo "TBN_WGRAD_LDS_PAD=$v $(TBN_WGRAD_LDS_PAD=$v step)"; done; done ;;
  branch4) for i in 1 2; do for v in none Audio all; do echo "branch=$v $(step --branch-streams $v)"; done; done ;;
  chunk) for c in 256 320 400 256 400; do
           echo "eval_chunk=$c $(python bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline --profile-steps 0 --eval-chunk $c 2>/dev/null | grep -o '"value": [0-9.]*')"; done ;;
  *) echo "usage: $0 lpt|prio|wpad|branch4|chunk"; exit 1 ;;
esac
