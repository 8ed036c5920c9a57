#!/bin/bash
# round 5, GPU call 1: new parity tests + baseline bench lines + kernel traces of the multi-stream step (critical-path attribution)
set -o pipefail
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_trainstep_gpu.py tests/test_operating_points_gpu.py "tests/test_model_gpu.py::test_config4_full_batch_train_step_vs_oracle" -q -m gpu -s -k "not config5" > $O/r05a_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -5 $O/r05a_pytest.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05a_bench_driver.json 2> $O/r05a_bench_driver.err || exit 1
head -c 300 $O/r05a_bench_driver.json; echo
timeout -k 10 200 python bench.py --config 2 --no-cpu-baseline > $O/r05a_bench_config2.json 2> /dev/null || exit 1
head -c 300 $O/r05a_bench_config2.json; echo
cd /tmp && export TMPDIR=/tmp
for c in 4 2; do
  rm -rf /tmp/tr$c
  timeout -k 10 240 rocprofv3 --kernel-trace -d /tmp/tr$c -o t --output-format csv -- python3 $ROOT/bench.py --config $c --steps 8 --warmup 4 --no-cpu-baseline --profile-steps 0 > $O/r05a_traced_bench_config$c.json 2> /dev/null || exit 1
  python3 $ROOT/scripts/step_timeline.py "$(find /tmp/tr$c -name '*kernel_trace.csv' | head -1)" 2 > $O/r05a_timeline_config$c.txt
  cat $O/r05a_timeline_config$c.txt
done
echo done
