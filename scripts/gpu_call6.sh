#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_conv_variants_gpu.py -q -m gpu --maxfail=10 > gpurun_out/c6_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -5 gpurun_out/c6_pytest.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 120 python scripts/stft_profile.py 2>/dev/null | tee gpurun_out/c6_stft_profile.txt
bash scripts/ab_layers.sh 2>&1 | tee gpurun_out/c6_ab.log
