#!/bin/bash
set -o pipefail
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
timeout -k 10 120 python scripts/debug/hook_vs_grad.py > $O/r05b_hook_debug.txt 2>&1; echo "hook debug rc=$?"; tail -30 $O/r05b_hook_debug.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trg
timeout -k 10 240 rocprofv3 --kernel-trace -d /tmp/trg -o t --output-format csv -- python3 $ROOT/scripts/graph_trace.py 4 6 > $O/r05b_graph_trace_config4.log 2>&1 || { tail -5 $O/r05b_graph_trace_config4.log; exit 1; }
tail -3 $O/r05b_graph_trace_config4.log
python3 $ROOT/scripts/step_timeline.py "$(find /tmp/trg -name '*kernel_trace.csv' | head -1)" 2 > $O/r05b_timeline_graph_config4.txt; cat $O/r05b_timeline_graph_config4.txt
cp "$(find /tmp/trg -name '*kernel_trace.csv' | head -1)" /tmp/kt.csv; python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/kt.csv')))
print(len(rows), 'kernels in trace')
PY
cd $ROOT
timeout -k 10 800 python -m pytest tests/test_trainstep_gpu.py tests/test_dp_gpu.py "tests/test_operating_points_gpu.py::test_config3_full_batch_train_step_vs_oracle" "tests/test_model_gpu.py::test_config4_full_batch_train_step_vs_oracle" -q -m gpu -s -k "not 2.1" > $O/r05b_pytest.log 2>&1
echo "pytest rc=$?"; tail -8 $O/r05b_pytest.log; grep "per-layer parity\|accumulated update" $O/r05b_pytest.log
echo done
