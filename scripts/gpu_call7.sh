#!/bin/bash
mkdir -p gpurun_out
ROOT=$PWD
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_conv_variants_gpu.py -q -m gpu --maxfail=10 > gpurun_out/c7_pytest.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/c7_pytest.log
if [ $rc -ge 124 ]; then exit $rc; fi
for m in "3 224 224"; do
  echo "table-build $(timeout -k 10 120 python scripts/layer_profile.py $m 96 2>/dev/null | tee gpurun_out/c7_lp_3.txt | grep 'total conv')"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $ROOT/gpurun_out/c7_counters_avail.txt 2>&1 || true
rm -rf /tmp/pmc7
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d /tmp/pmc7 -o t --output-format csv -- python3 $ROOT/scripts/layer_profile.py 3 224 224 96 > /dev/null 2>&1
python3 $ROOT/scripts/pmc_sq.py /tmp/pmc7/t_counter_collection.csv 60 > $ROOT/gpurun_out/c7_pmc_wait_rgb.txt
head -5 $ROOT/gpurun_out/c7_pmc_wait_rgb.txt | cut -c1-250
