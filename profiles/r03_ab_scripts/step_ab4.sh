#!/bin/bash
# One call: forward recurrence hand-off by flags (default) vs tagged data (mode bit 29) in the fp32 training step.
cd "$(dirname "$0")/../.."
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['mean_loss'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})"; }
for rep in 1 2 3; do
  echo "== flags"; run
  echo "== tagged, hold-back 0.8"; SEPKERN_LSTM_FWD=0,1,1,0,0,8,0,1 run
  echo "== tagged, no hold-back"; SEPKERN_LSTM_FWD=0,1,1,0,0,31,0,1 run
done
