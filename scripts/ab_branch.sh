#!/bin/bash
# same-box A/B of the branch-level side stream (bench.py --branch-streams): alternating runs, ms per step
# usage: bash scripts/ab_branch.sh <config> <armA> <armB> [reps]     e.g.  2 none RGB     4 none all
CFG=${1:-2}; A=${2:-none}; B=${3:-all}; REPS=${4:-3}; EXTRA=${5:-}
for i in $(seq $REPS); do
  for v in $A $B; do
    echo "config $CFG branch=$v $EXTRA $(python bench.py --config $CFG --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 0 --branch-streams $v $EXTRA 2>/dev/null | grep -o 'ms_per_step.: [0-9.]*')"
  done
done
