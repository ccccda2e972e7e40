#!/bin/bash
# TIMING-ONLY upper bound (wrong but finite numerics) of "operands that arrive split": planesfree = the 256 x 128 kernel's
# staging split costs nothing (-DSK_PLANES_FREE, a transient edit of split4, not in the tree); allfree = that + the T/N form of the
# 128 x 128 kernel with its pieces for free (-DSK_SPLIT_FREE_TN): every large product of the step without split VALU work.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_all_split_free.txt
: > $OUT
cd $R
for i in 1 2; do
  for name in default planesfree allfree; do
    lib=$R/speech-separation_amd/sepkern/libsepkern.so; [ $name != default ] && lib=$R/speech-separation_amd/sepkern/libsepkern_$name.so
    SEPKERN_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 2>&1 | python -c "
import json,sys
l=[x for x in sys.stdin.read().strip().splitlines() if x.startswith('{')]
if not l: print('$name: no line'); sys.exit(0)
d=json.loads(l[-1]); k=d['kernels']
print('%-10s %.3f ms/step  ' % ('$name', d['ms_per_step']) + '  '.join('%s %.2f' % (n, v['ms_per_step']) for n, v in k.items()))
" | tee -a $OUT
  done
done
