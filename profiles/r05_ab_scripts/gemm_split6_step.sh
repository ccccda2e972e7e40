#!/bin/bash
# (1) stand-alone: the shipped fp32-MFMA kernels (variant 0) vs the 128 x 128 split kernel (variant 2) vs the 256 x 256 stream-K kernel with
# split products (variant 7), all six-product, pieces by rounding; (2) the training step with the split kernels on the main stream
# and / or beside the recurrences: SEPKERN_GEMM_VARIANTS=main,side.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_gemm_split6_step.txt
T=$R/speech-separation_amd/tools
: > $O
S="--shape 12800,7168,1792,0,1 --shape 12800,1792,7168,0,0 --shape 7168,1792,12800,1,0 --shape 12800,514,1792,0,1"
for v in 0 2 7 6; do echo "== variant $v" >> $O; python3 $T/gemm_bench.py --variant $v $S >> $O 2>/dev/null || exit 1; done
for i in 1 2; do
  for gv in 0,1 2,1 7,1 2,2 7,2; do
    SEPKERN_GEMM_VARIANTS=$gv python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2> /dev/null |
      python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('variants $gv run $i: %.3f ms/step  %.0f frames/s  ' % (d['ms_per_step'], d['value']) + '  '.join('%s %.2f' % (n, v['ms_per_step']) for n, v in sorted(k.items()) if v['ms_per_step'] > 1.0) + '  loss %.5f' % d['config']['mean_loss'])" >> $O || exit 1
  done
done
cat $O
