#!/bin/bash
# layer-0 input width padded to a multiple of 16 (257 -> 272: the layer-0 products take the split kernels) against 4 (260: fp32-MFMA kernels)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_pad_in_step.txt
: > $O
for BARGS in "" "--arch rsh --hidden 600 --layers 2 --num-spk 4"; do
  for i in 1 2 3; do
    for pad in 4 16; do
      SEPKERN_PAD_IN=$pad python3 $R/bench.py $BARGS --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2> /dev/null |
        python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('%-22s pad=$pad run $i: %.3f ms/step  %.0f frames/s  ' % ('$BARGS' or 'uniform', d['ms_per_step'], d['value']) + '  '.join('%s %.2f' % (n, v['ms_per_step']) for n, v in sorted(k.items()) if v['ms_per_step'] > 0.5) + '  loss %.5f' % d['config']['mean_loss'])" >> $O || exit 1
    done
  done
done
cat $O
