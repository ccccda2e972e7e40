#!/bin/bash
# Backward recurrence: geometry of the per-wave DMA ring (sub-blocks per step, sub-blocks in flight), one call.
cd "$(dirname "$0")/.."
for rep in 1 2; do
for lib in "" r7_3 r14_4 r14_5 r4_2; do
  if [ -n "$lib" ]; then export SEPKERN_LIB=$PWD/sepkern/libsepkern_$lib.so; else unset SEPKERN_LIB; fi
  echo "== ${lib:-default r8_3}"
  timeout -k 10 200 python tools/lstm_bench.py --rounds 5 --fwd "0,1,1" --bwd "0,1,0,0,0,0,31" 2>&1 | grep " bwd "
done
done
