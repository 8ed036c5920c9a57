#!/bin/bash
# Timing diagnostic (results invalid): what the forward / backward BN finalize launches of the batched BN steps cost the
# three-stream step.  TBN_DIAG_SKIP: 16 forward, 32 backward, 48 both.  The TBN_DIAG=1 library is built beside the shipped
# one (scripts/ab/lib_diag.so) and selected with TBN_LIB: libtbn_hip.so is never touched.
set -eo pipefail
TBN_BUILD_VARIANT=diag TBN_DIAG=1 python -m attention_based_tbn_amd.build > /dev/null
export TBN_LIB=$PWD/scripts/ab/lib_diag.so
for i in 1 2 3; do
  for m in 0 16 32 48; do
    echo "TBN_DIAG_SKIP=$m $(TBN_DIAG_SKIP=$m python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 0 2>/dev/null | grep -o 'ms_per_step.: [0-9.]*')"
  done
done
