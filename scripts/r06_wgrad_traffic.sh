#!/bin/bash
set -eo pipefail
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_wg
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum -d /tmp/pmc_wg -o t --output-format csv -- python3 $ROOT/scripts/wgrad_traffic.py > /tmp/wg.log 2>&1 || { tail -20 /tmp/wg.log; exit 1; }
python3 $ROOT/scripts/wgrad_traffic.py "$(find /tmp/pmc_wg -name '*counter_collection.csv' | head -1)" | tee $ROOT/gpurun_out/r06u_wgrad_traffic.txt
