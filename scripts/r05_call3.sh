#!/bin/bash
set -o pipefail
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu -x -k "not full_batch and not config5" > $O/r05c_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -6 $O/r05c_pytest.log
if [ $rc -ge 124 ]; then exit $rc; fi
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 0"
ms() { grep -o 'ms_per_step.: [0-9.]*' | head -1; }
echo "== config 4: riders off / on (alternating)"
for rep in 1 2 3; do
  for r in off on; do echo "riders=$r $($B --riders $r 2>/dev/null | ms)"; done
done | tee $O/r05c_ab_riders_config4.txt
echo "== config 4: riders on, in front of the GEMM tiles"
for rep in 1 2; do echo "front $(TBN_RIDER_FRONT=1 $B --riders on 2>/dev/null | ms)"; done | tee -a $O/r05c_ab_riders_config4.txt
echo "== config 2"
for rep in 1 2 3; do
  echo "branch+aux(shipped r04) $($B --config 2 --riders off 2>/dev/null | ms)"
  echo "riders+aux, no branch   $($B --config 2 --riders on --branch-streams none 2>/dev/null | ms)"
  echo "riders only             $($B --config 2 --riders on --branch-streams none --no-aux-stream 2>/dev/null | ms)"
  echo "nothing                 $($B --config 2 --riders off --branch-streams none --no-aux-stream 2>/dev/null | ms)"
done | tee $O/r05c_ab_riders_config2.txt
for c in 4 2; do
  for r in off on; do
    extra=""; if [ $c = 2 ] && [ $r = on ]; then extra="--branch-streams none"; fi
    python bench.py --config $c --steps 10 --warmup 5 --no-cpu-baseline --profile-steps 0 --riders $r $extra --timeline $O/r05c_tl_config${c}_riders_$r.csv > /dev/null 2>&1 || exit 1
    python scripts/step_timeline.py $O/r05c_tl_config${c}_riders_$r.csv 2 > $O/r05c_timeline_config${c}_riders_$r.txt
    echo "---- config $c riders $r"; cat $O/r05c_timeline_config${c}_riders_$r.txt
    rm -f $O/r05c_tl_config${c}_riders_$r.csv
  done
done
echo done
