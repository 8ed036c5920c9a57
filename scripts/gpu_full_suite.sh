#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu --maxfail=10 > gpurun_out/c3_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a gpurun_out/c3_pytest.log
tail -15 gpurun_out/c3_pytest.log
