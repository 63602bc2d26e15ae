#!/bin/bash
# Build libuvit.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=${1:-../libuvit.so}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffast-math -fno-finite-math-only -Wall -Wno-unused-function"
# Fingerprint of the sources this library is built from: native.lib() recomputes it and refuses a stale .so
# (struct layouts in include/uvit.h travel by value through ctypes).
HASH=$(cat $(ls *.hip *.h | LC_ALL=C sort) ../../include/uvit.h | sha256sum | cut -c1-16)
FLAGS="$FLAGS -DUVIT_SRC_HASH=\"$HASH\" $UVIT_EXTRA_FLAGS"   # UVIT_EXTRA_FLAGS: -D switches of A/B experiments
mkdir -p obj
pids=()
for f in gemm attention attention2 norm elementwise optim engine; do
  ( hipcc $FLAGS -c $f.hip -o obj/$f.o ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
# Build-time guard (round 4): no taken branch between an MFMA and the first read of its result without the wait states the MFMA
# needs -- hipcc pads the fall-through path only (tools/check_mfma_hazard.py; tools/micro/mfma_branch_hazard.hip is the flagged case).
if [ -z "$UVIT_SKIP_HAZARD_CHECK" ]; then
  python3 ../../tools/check_mfma_hazard.py --compile gemm.hip attention.hip attention2.hip $UVIT_EXTRA_FLAGS || { echo "MFMA hazard check FAILED"; exit 1; }
fi
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" obj/*.o
echo "$HASH" > "$OUT.hash"
echo "built $OUT"
