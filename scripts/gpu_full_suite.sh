#!/bin/bash
# the whole `-m gpu` suite into gpurun_out/<tag>_pytest_full.log (what a round-end check runs), slowest tests listed
TAG=${1:-suite}
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -q -m gpu --maxfail=10 --durations=25 > gpurun_out/${TAG}_pytest_full.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a gpurun_out/${TAG}_pytest_full.log
tail -n 45 gpurun_out/${TAG}_pytest_full.log
