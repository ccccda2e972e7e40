#!/bin/bash
# TIMING-ONLY upper bound (wrong but finite numerics, -DSK_SPLIT_FREE_TN): the T/N form of the 128 x 128 split kernel -- the weight
# gradients, i.e. every product that runs BESIDE a backward recurrence plus layer 0's on the main stream -- finds its bf16 pieces
# for free (no VALU split work; same DMA, fragment reads and MFMAs): what a split-once kernel small enough to be hosted could gain
# at best.  Two alternations.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_hosted_split_free.txt
: > $OUT
cd $R
for i in 1 2; do
  for name in default freetn; do
    lib=$R/speech-separation_amd/sepkern/libsepkern.so; [ $name = freetn ] && lib=$R/speech-separation_amd/sepkern/libsepkern_freetn.so
    SEPKERN_LIB=$lib python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 2>&1 | python -c "
import json,sys
l=[x for x in sys.stdin.read().strip().splitlines() if x.startswith('{')]
if not l: print('$name: no line'); sys.exit(0)
d=json.loads(l[-1]); k=d['kernels']
print('%-8s %.3f ms/step  ' % ('$name', d['ms_per_step']) + '  '.join('%s %.2f' % (n, v['ms_per_step']) for n, v in k.items()))
" | tee -a $OUT
  done
done
