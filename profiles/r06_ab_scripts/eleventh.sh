#!/bin/bash
# r06, eleventh GPU call (bf16 3-spk): K slices of the weight-gradient launches beside the XCD-local backward recurrences -- the
# library's choice (split-K + a reduce launch that then runs on the 32 CUs the recurrence leaves free) against one slice.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
OUT=$O/r06_bf16_side_splitk.txt
: > $OUT
line() { python -c "
import json,sys
l=[x for x in sys.stdin.read().strip().splitlines() if x.startswith('{')]
if not l: print('$1: no line'); sys.exit(0)
d=json.loads(l[-1]); k=d['kernels']
print('%-12s %.3f ms/step  %.0f frames/s  loss %.6f  ' % ('$1', d['ms_per_step'], d['value'], d['config']['mean_loss']) + '  '.join('%s %.2f' % (n, v['ms_per_step']) for n, v in k.items()))
"; }
for i in 1 2 3; do
  for v in 0 1 2; do
    SEPKERN_BF16_SIDE_SPLITK=$v timeout -k 10 200 python bench.py --dtype bf16 --num-spk 3 --no-cpu-baseline --no-secondary --no-aux --no-power-probe --steps 20 --warmup 3 2>/dev/null | line splitk_$v | tee -a $OUT
  done
done
for v in 0 1; do
  SEPKERN_BF16_SIDE_SPLITK=$v timeout -k 10 200 python bench.py --dtype bf16 --ragged --no-cpu-baseline --no-secondary --no-aux --no-power-probe --steps 20 --warmup 3 2>/dev/null | line ragged_$v | tee -a $OUT
done
