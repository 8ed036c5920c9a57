#!/bin/bash
set -o pipefail
ROOT=$PWD
O=$ROOT/gpurun_out
mkdir -p $O
timeout -k 10 200 python scripts/debug/hook_vs_grad.py > $O/r05e_hook_debug.txt 2>&1; echo "debug rc=$?"; grep -v "Freezing\|amdgpu.ids" $O/r05e_hook_debug.txt | tail -40
timeout -k 10 900 python -m pytest tests/test_dp_gpu.py tests/test_trainstep_gpu.py tests/test_kernels_gpu.py tests/test_riders_gpu.py tests/test_model_gpu.py -q -m gpu -k "not full_batch" > $O/r05e_pytest.log 2>&1
echo "pytest rc=$?"; tail -8 $O/r05e_pytest.log
echo done
