#!/bin/bash
# the training step with the split-once-while-staging kernel chosen for the large unsplit main-stream products (default) against
# SEPKERN_GEMM_PLANES=0 (the 128 x 128 / stream-K split kernels there), two alternations, three workloads
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_gemm_planes_step.txt
: > $O
for BARGS in "" "--ragged" "--arch rsh --hidden 600 --layers 2 --num-spk 4"; do
  for i in 1 2 3; do
    for pl in 0 1; do
      SEPKERN_GEMM_PLANES=$pl python3 $R/bench.py $BARGS --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2> /dev/null |
        python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('%-22s planes=$pl run $i: %.3f ms/step  %.0f frames/s  ' % ('$BARGS' or 'uniform', d['ms_per_step'], d['value']) + '  '.join('%s %.2f' % (n, v['ms_per_step']) for n, v in sorted(k.items()) if v['ms_per_step'] > 1.0) + '  loss %.5f' % d['config']['mean_loss'])" >> $O || exit 1
    done
  done
done
cat $O
