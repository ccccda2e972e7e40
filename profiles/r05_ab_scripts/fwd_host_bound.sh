#!/bin/bash
# Upper bound of hosting the next layer's input projection beside the forward recurrence (timing-only: SEPKERN_ABL_FWD_HOST=1
# launches that projection on the side stream reading y BEFORE it is written; results wrong by construction).  Alternating runs.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_fwd_host_bound.txt
: > $OUT
cd $R
for i in 1 2 3; do
  for abl in 0 1; do
    SEPKERN_ABL_FWD_HOST=$abl python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
k=d['kernels']
print('abl=$abl  %.3f ms/step   ' % d['ms_per_step'] + '  '.join('%s %.2f' % (n, v['ms_per_step']) for n, v in k.items()))
" | tee -a $OUT
  done
done
for abl in 0 1; do
  SEPKERN_ABL_FWD_HOST=$abl python bench.py --no-cpu-baseline --no-secondary --ragged --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('ragged abl=$abl  %.3f ms/step' % d['ms_per_step'])
" | tee -a $OUT
done
