#!/bin/bash
# Timing diagnostic (results invalid): what the forward / backward BN finalize launches of the batched BN steps cost the
# three-stream step.  Needs scripts/ab/lib_diag.so = a TBN_DIAG=1 build.  TBN_DIAG_SKIP: 16 forward, 32 backward, 48 both.
L=attention_based_tbn_amd/libtbn_hip.so
cp $L /tmp/orig_fs.so
cp scripts/ab/lib_diag.so $L
for i in 1 2 3; do
  for m in 0 16 32 48; do
    echo "TBN_DIAG_SKIP=$m $(TBN_DIAG_SKIP=$m python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 0 2>/dev/null | grep -o 'ms_per_step.: [0-9.]*')"
  done
done
cp /tmp/orig_fs.so $L
