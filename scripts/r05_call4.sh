#!/bin/bash
set -o pipefail
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu -k "not full_batch and not config5" > $O/r05d_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -12 $O/r05d_pytest.log
if [ $rc -ge 124 ]; then exit $rc; fi
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 0"
ms() { grep -o 'ms_per_step.: [0-9.]*' | head -1; }
for rep in 1 2 3; do echo "config4 $($B 2>/dev/null | ms)   config2 $($B --config 2 2>/dev/null | ms)   config3 $($B --config 3 2>/dev/null | ms)"; done | tee $O/r05d_bench_repeats.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trt
timeout -k 10 240 rocprofv3 --kernel-trace -d /tmp/trt -o t --output-format csv -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --profile-steps 0 > /dev/null 2>&1 || exit 1
python3 $ROOT/scripts/turn_timeline.py "$(find /tmp/trt -name '*kernel_trace.csv' | head -1)" > $O/r05d_turn_timeline_traced.txt; cat $O/r05d_turn_timeline_traced.txt
echo done
