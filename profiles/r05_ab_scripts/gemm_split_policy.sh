#!/bin/bash
# the training step under the new default (split kernels chosen by the library on both streams) with the stream-K form of the split
# kernel never / for long-K N/N + T/N products / also for the N/T projections, against the r04 arrangement (8,1), on the uniform
# and the ragged workload and RSH.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_gemm_split_policy.txt
: > $O
run() { # label, env..., -- bench args
  local label=$1; shift
  env "$@" python3 $R/bench.py $BARGS --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2> /dev/null |
    python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('%-22s %-10s %.3f ms/step  %.0f frames/s  ' % ('$BARGS' or 'uniform', '$label', d['ms_per_step'], d['value']) + '  '.join('%s %.2f' % (n, v['ms_per_step']) for n, v in sorted(k.items()) if v['ms_per_step'] > 1.0) + '  loss %.5f' % d['config']['mean_loss'])" >> $O || exit 1
}
for BARGS in "" "--ragged" "--arch rsh --hidden 600 --layers 2 --num-spk 4"; do
  for i in 1 2; do
    run r04-8,1 SEPKERN_GEMM_VARIANTS=8,1
    run sk0 SEPKERN_GEMM_SPLIT_SK=0
    run sk1 SEPKERN_GEMM_SPLIT_SK=1
    run sk2 SEPKERN_GEMM_SPLIT_SK=2
  done
done
cat $O
