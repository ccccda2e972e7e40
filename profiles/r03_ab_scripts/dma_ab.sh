#!/bin/bash
# A/B in ONE call: LDS-DMA of the recurrence kernels as inline assembly (shipped) vs the builtin (libsepkern_dmab.so).
cd "$(dirname "$0")/.."
for lib in "" dmab "" dmab; do
  echo "== lib ${lib:-default (asm DMA)}"
  if [ -n "$lib" ]; then export SEPKERN_LIB=$PWD/sepkern/libsepkern_$lib.so; else unset SEPKERN_LIB; fi
  timeout -k 10 200 python tools/lstm_bench.py --rounds 5 --fwd "0,1,1" --bwd "0,1,0,0,0,0,31" 2>&1 | grep -v amdgpu.ids
  timeout -k 10 200 python tools/lstm_bench.py --bf16 --rounds 5 --fwd "0,1,1,0,0,1" --bwd "0,1,0,0,0,0,31" 2>&1 | grep -v amdgpu.ids
done
cd ..
for lib in "" dmab "" dmab; do
  if [ -n "$lib" ]; then export SEPKERN_LIB=$PWD/speech-separation_amd/sepkern/libsepkern_$lib.so; else unset SEPKERN_LIB; fi
  echo "== bench lib ${lib:-default (asm DMA)}"
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})"
done
