#!/bin/bash
# forward recurrence, split product: two gate-row tiles over a quarter of K per wave (P2, the tree) against one tile over half of K
# (_ab/libsepkern_p1.so, the tree before): tests, then the step, three alternations, uniform + ragged + RSH
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_step_ab_fwd_p2.txt
: > $O
for BARGS in "" "--ragged" "--arch rsh --hidden 600 --layers 2 --num-spk 4"; do
  for i in 1 2 3; do
    for v in p1 p2; do
      if [ $v = p1 ]; then export SEPKERN_LIB=$R/_ab/libsepkern_p1.so; else unset SEPKERN_LIB; fi
      python3 $R/bench.py $BARGS --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2> /dev/null |
        python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); by=d['roofline']['by_kernel']
print('%-22s $v run $i: %.3f ms/step  %.0f frames/s   fwd rec %.3f ms (%.2f us/step)  bwd rec %.3f ms  loss %.6f' % ('$BARGS' or 'uniform', d['ms_per_step'], d['value'], by['lstm_fwd_kernel']['ms_per_step'], by['lstm_fwd_kernel']['us_per_time_step'], by['lstm_bwd_kernel']['ms_per_step'], d['config']['mean_loss']))" >> $O || exit 1
    done
  done
done
unset SEPKERN_LIB
cat $O
