#!/bin/bash
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 0"
ms() { grep -o 'ms_per_step.: [0-9.]*' | head -1; }
for rep in 1 2; do
  echo "cfg4 no aux (shipped)            $($B 2>/dev/null | ms)"
  for g in 1 4 8 16; do
    echo "cfg4 aux all, group $g            $(TBN_AUX_GROUP=$g $B --aux-streams RGB,Flow,Audio 2>/dev/null | ms)"
  done
  echo "cfg4 aux Audio only, group 8     $(TBN_AUX_GROUP=8 $B --aux-streams Audio 2>/dev/null | ms)"
done
for rep in 1 2; do
  echo "cfg2 branch+aux (shipped)        $($B --config 2 2>/dev/null | ms)"
  for g in 1 4 8 16; do
    echo "cfg2 aux, no branch, group $g     $(TBN_AUX_GROUP=$g $B --config 2 --branch-streams none 2>/dev/null | ms)"
  done
done
for g in 1 8; do
  echo "cfg3 aux both, group $g           $(TBN_AUX_GROUP=$g $B --config 3 --aux-streams RGB,Audio 2>/dev/null | ms)"
done
echo "cfg3 no aux (shipped)            $($B --config 3 2>/dev/null | ms)"
