#!/bin/bash
# Hold-back of a step's first poll (SEPKERN_LSTM_FWD / _BWD 6th field, x 0.1 us; 0 = the library's choice = 0.8 us forward on a
# full grid, 31 = none) re-swept on the r05 kernels (read rings, six products) in the training step.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_poll_delay_sweep.txt
: > $OUT
cd $R
run() {
  SEPKERN_LSTM_FWD=$2 SEPKERN_LSTM_BWD=$3 timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
l=[x for x in sys.stdin.read().strip().splitlines() if x.startswith('{')]
if not l: print('$1: no line'); sys.exit(0)
d=json.loads(l[-1]); k=d['kernels']
print('%-18s %.3f ms/step  fwd %.2f  bwd %.2f  side %.2f' % ('$1', d['ms_per_step'], k['lstm_fwd_kernel']['ms_per_step'], k['lstm_bwd_kernel']['ms_per_step'], k['gemm_f32_split_kernel@side']['ms_per_step']))
" | tee -a $OUT
}
for i in 1 2; do
  run "fwd 8 (default)" 0,1,1,0,0,0,0,1  0,1,0,0,0,31,0,0
  run "fwd 4"           0,1,1,0,0,4,0,1  0,1,0,0,0,31,0,0
  run "fwd 6"           0,1,1,0,0,6,0,1  0,1,0,0,0,31,0,0
  run "fwd 10"          0,1,1,0,0,10,0,1 0,1,0,0,0,31,0,0
  run "fwd none"        0,1,1,0,0,31,0,1 0,1,0,0,0,31,0,0
  run "bwd 4"           0,1,1,0,0,0,0,1  0,1,0,0,0,4,0,0
  run "bwd 8"           0,1,1,0,0,0,0,1  0,1,0,0,0,8,0,0
done
