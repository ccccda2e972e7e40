#!/bin/bash
# Step-level A/B of the planes kernel builds: noflip (the r05h kernel), flip32 without the stated interleave, and the stated
# interleave with 3 / 4 VALU instructions behind every MFMA (both with flip32).  Three alternations, uniform 32 x 400; one on the ragged set.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_planes_sched_step.txt
: > $OUT
lib_of() { if [ $1 = flip32 ]; then echo $R/speech-separation_amd/sepkern/libsepkern.so; else echo $R/speech-separation_amd/sepkern/libsepkern_$1.so; fi; }
cd $R
for i in 1 2 3; do
  for name in noflip flip32 sched3 sched4; do
    SEPKERN_LIB=$(lib_of $name) python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
k=d['kernels']
print('%-7s %.3f ms/step   ' % ('$name', d['ms_per_step']) + '  '.join('%s %.2f' % (n, v['ms_per_step']) for n, v in k.items()))
" | tee -a $OUT
  done
done
for name in noflip sched3 sched4; do
  SEPKERN_LIB=$(lib_of $name) python bench.py --no-cpu-baseline --no-secondary --ragged --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('ragged %-7s %.3f ms/step  %.0f valid frames/s' % ('$name', d['ms_per_step'], d['value']))
" | tee -a $OUT
done
