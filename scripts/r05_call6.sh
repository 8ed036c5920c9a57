#!/bin/bash
# round 5: the full GPU suite as the driver runs it + rider A/B and library timelines kept as evidence
set -o pipefail
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > $O/r05f_pytest_full.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -6 $O/r05f_pytest_full.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 0"
ms() { grep -o 'ms_per_step.: [0-9.]*' | head -1; }
{
echo "# same box, alternating; bench.py --steps 40 --warmup 5 --profile-steps 0"
for rep in 1 2 3; do
  for r in off on; do echo "config 4 riders=$r $($B --riders $r 2>/dev/null | ms)"; done
done
for rep in 1 2; do
  for r in off on; do echo "config 3 riders=$r $($B --config 3 --riders $r 2>/dev/null | ms)"; done
done
for rep in 1 2 3; do
  echo "config 2 branch+aux (shipped policy)  $($B --config 2 --riders off 2>/dev/null | ms)"
  echo "config 2 riders+aux, no branch        $($B --config 2 --riders on --branch-streams none 2>/dev/null | ms)"
  echo "config 2 riders only                  $($B --config 2 --riders on --branch-streams none --no-aux-stream 2>/dev/null | ms)"
  echo "config 2 one chain, no riders         $($B --config 2 --riders off --branch-streams none --no-aux-stream 2>/dev/null | ms)"
done
} | tee $O/r05_ab_riders.txt
for c in 4 2; do
  for r in off on; do
    extra=""; if [ $c = 2 ] && [ $r = on ]; then extra="--branch-streams none"; fi
    python bench.py --config $c --steps 10 --warmup 5 --no-cpu-baseline --profile-steps 0 --riders $r $extra --timeline /tmp/tl.csv > /dev/null 2>&1 || exit 1
    { echo "# bench.py --config $c --riders $r $extra --timeline (library kernel timeline: no external tracer)"; python scripts/step_timeline.py /tmp/tl.csv 2; } > $O/r05_timeline_config${c}_riders_$r.txt
    echo "---- config $c riders $r"; head -4 $O/r05_timeline_config${c}_riders_$r.txt
  done
done
echo done
