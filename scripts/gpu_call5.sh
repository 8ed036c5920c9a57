#!/bin/bash
# STFT parity + profile; graph-capture experiment under rocgdb (MULTI=0 AUX=1 first: never recorded; then MULTI=1 AUX=1)
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "stft or mel" > gpurun_out/c5_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/c5_pytest.log
timeout -k 10 120 python scripts/stft_profile.py 2>/dev/null | tee gpurun_out/c5_stft_profile.txt
for cfg in "0 1" "1 1"; do
  set -- $cfg
  echo "=== MULTI=$1 AUX=$2 ===" | tee -a gpurun_out/c5_graph.log
  MULTI=$1 AUX=$2 timeout -k 10 300 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGSEGV stop nopass" -ex run -ex "bt 40" -ex "info sharedlibrary amdhip" -ex "thread apply all bt 14" --args python3 scripts/graph_experiment.py > gpurun_out/c5_graph_M$1_A$2.log 2>&1
  echo "rc=$?" | tee -a gpurun_out/c5_graph.log
  grep -n "eager\|captured\|graph replay\|capture\]\|SIGSEGV\|^#" gpurun_out/c5_graph_M$1_A$2.log | tail -60 | tee -a gpurun_out/c5_graph.log
done
