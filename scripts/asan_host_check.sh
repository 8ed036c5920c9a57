#!/bin/bash
# Host-side AddressSanitizer pass over the C-ABI library (CPU only: GPU ASAN is not available on this pool).
# Builds every csrc/*.hip with the HOST half instrumented (-Xarch_host -fsanitize=address) into /tmp/tbn_asan and
# runs the CPU test-suite's library users (plan creation / validation, ABI argument rejection, symbol export)
# against that build.  Nothing in the repo is replaced.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=/tmp/tbn_asan
RT=$(find /opt/rocm/lib/llvm -name 'libclang_rt.asan-x86_64.so' | head -1)
mkdir -p "$OUT"
for f in "$ROOT"/attention_based_tbn_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -fPIC -std=c++17 -Xarch_host -fsanitize=address \
    -Xarch_host -fno-omit-frame-pointer -I"$ROOT/include" -c "$f" -o "$OUT/$(basename "${f%.hip}").o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -shared-libasan -o "$OUT/libtbn_hip.so" "$OUT"/*.o
cd "$ROOT"
LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0 python -c "
import attention_based_tbn_amd._lib as L
L.LIB_PATH = '$OUT/libtbn_hip.so'
import pytest, sys
sys.exit(pytest.main(['tests/test_host_cpu.py', '-x', '-q', '-m', 'not gpu', '-p', 'no:cacheprovider']))
"
