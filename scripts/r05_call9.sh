#!/bin/bash
B="python bench.py --config 3 --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 0"
ms() { grep -o 'ms_per_step.: [0-9.]*' | head -1; }
for rep in 1 2 3; do
  echo "default                      $($B 2>/dev/null | ms)"
  echo "aux Audio                    $($B --aux-streams Audio 2>/dev/null | ms)"
  echo "aux Audio + branch Audio     $($B --aux-streams Audio --branch-streams Audio 2>/dev/null | ms)"
  echo "aux both                     $($B --aux-streams RGB,Audio 2>/dev/null | ms)"
  echo "riders off                   $($B --riders off 2>/dev/null | ms)"
done
