#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu --maxfail=10 > gpurun_out/c8_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -4 gpurun_out/c8_pytest.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 120 python scripts/stft_profile.py 2>/dev/null | tee gpurun_out/c8_stft_profile.txt
bash scripts/ab_layers.sh 2>&1 | tee gpurun_out/c8_ab.log
