#!/bin/bash
# A/B, one gpurun call, three alternations: the forward recurrence's split product with all NINE piece products (r04: _ab/libsepkern_s9.so,
# the tree before this change) vs the SIX that reach fp32's resolution (the shipped library), on the headline shape and on the ragged set.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_step_ab_fwd_six_products.txt
: > $O
for wl in "" "--ragged"; do
  for i in 1 2 3; do
    for v in nine six; do
      if [ $v = nine ]; then export SEPKERN_LIB=$R/_ab/libsepkern_s9.so; else unset SEPKERN_LIB; fi
      python3 $R/bench.py $wl --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2> /dev/null |
        python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); by=d['roofline']['by_kernel']
print('%-8s $v run $i: %.3f ms/step  %.0f frames/s   fwd rec %.3f ms (%.2f us/step)  bwd rec %.3f ms  mean loss %.6f' % ('${wl:-uniform}', d['ms_per_step'], d['value'], by['lstm_fwd_kernel']['ms_per_step'], by['lstm_fwd_kernel']['us_per_time_step'], by['lstm_bwd_kernel']['ms_per_step'], d['config']['mean_loss']))" >> $O || exit 1
    done
  done
done
unset SEPKERN_LIB
cat $O
