#!/bin/bash
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 0"
ms() { grep -o 'ms_per_step.: [0-9.]*' | head -1; }
for rep in 1 2 3; do
  echo "three streams (shipped)         $($B 2>/dev/null | ms)"
  echo "Flow on RGB's stream            $(TBN_SHARE_STREAM=Flow:RGB $B 2>/dev/null | ms)"
  echo "Flow on Audio's stream          $(TBN_SHARE_STREAM=Flow:Audio $B 2>/dev/null | ms)"
  echo "RGB on Audio's stream           $(TBN_SHARE_STREAM=RGB:Audio $B 2>/dev/null | ms)"
done
