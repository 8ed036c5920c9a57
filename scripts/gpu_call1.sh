#!/bin/bash
# round-3 call 1: parity of the new wgrad addressing / VGPR-form build, then same-box A/B against the round-2 library
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -x -q -m gpu > gpurun_out/c1_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a gpurun_out/c1_pytest.log
tail -5 gpurun_out/c1_pytest.log
if [ $rc -ge 124 ]; then exit $rc; fi     # killed at its limit: no further GPU step in this call
bash scripts/ab_layers.sh 2>&1 | tee gpurun_out/c1_ab.log
