#!/bin/bash
# cost side of folding the BN apply of a 3x3 layer's input into its consumers (diagnostic build, results unchanged):
# LDS-halo forward staging + weight-gradient x staging apply scale / shift / ReLU; same box, alternating.  The TBN_DIAG=1
# library is built beside the shipped one and selected with TBN_LIB.
set -o pipefail
mkdir -p gpurun_out
TBN_BUILD_VARIANT=diag TBN_DIAG=1 python -m attention_based_tbn_amd.build > /dev/null || exit 1
export TBN_LIB=$PWD/scripts/ab/lib_diag.so
timeout -k 10 300 env TBN_DIAG_FOLD=1 TBN_FORCE_HALO=1 python -m pytest tests/test_model_gpu.py -q -m gpu -k "train_step_matches" > gpurun_out/c16_pytest.log 2>&1
echo "parity with the fold probe on: rc=$?"; tail -2 gpurun_out/c16_pytest.log
for rep in 1 2 3; do
  for f in 0 1; do
    echo "fold=$f $(TBN_DIAG_FOLD=$f TBN_FORCE_HALO=1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-every 0 2>/dev/null | grep -o 'ms_per_step.: [0-9.]*')"
  done
done
for f in 0 1; do
  TBN_DIAG_FOLD=$f TBN_FORCE_HALO=1 timeout -k 10 120 python scripts/layer_profile.py 3 224 224 96 2>/dev/null > gpurun_out/c16_lp_fold$f.txt
  echo "fold=$f $(grep 'total conv' gpurun_out/c16_lp_fold$f.txt)"
done
