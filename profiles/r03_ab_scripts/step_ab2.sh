#!/bin/bash
# One call, one device: schedule switches of the fp32 step on the r03 kernels.
cd "$(dirname "$0")/../.."
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})"; }
for rep in 1 2; do
  echo "== default"; run
  echo "== SEPKERN_FWD_SPLIT=1"; SEPKERN_FWD_SPLIT=1 run
  echo "== SEPKERN_BWD_SPLIT=1"; SEPKERN_BWD_SPLIT=1 run
  echo "== ring 14x4 (libsepkern_r14_4.so)"; SEPKERN_LIB=$PWD/speech-separation_amd/sepkern/libsepkern_r14_4.so run
  echo "== SEPKERN_OVERLAP=0"; SEPKERN_OVERLAP=0 run
  echo "== SEPKERN_LSTM_FWD=0,1,1,0,0,16,1 (two-stream forward, hold-back 1.6 us)"; SEPKERN_LSTM_FWD=0,1,1,0,0,16,1 run
done
