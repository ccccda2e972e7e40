#!/bin/bash
# A/B of the fragment-read ring in the split forward recurrence's product (SK_FWD_RING = reads in flight ahead of their products):
# training step (three alternations) and the recurrence parity tests on the ring builds.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_fwd_ring.txt
: > $OUT
lib_of() { if [ $1 = ring0 ]; then echo $R/speech-separation_amd/sepkern/libsepkern.so; else echo $R/speech-separation_amd/sepkern/libsepkern_$1.so; fi; }
cd $R
for i in 1 2 3; do
  for name in ring0 ring2 ring3; do
    SEPKERN_LIB=$(lib_of $name) python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
k=d['kernels']
print('%-6s %.3f ms/step  loss %.6f  ' % ('$name', d['ms_per_step'], d['config']['mean_loss']) + '  '.join('%s %.2f' % (n, v['ms_per_step']) for n, v in k.items()))
" | tee -a $OUT
  done
done
for name in ring0 ring3; do
  SEPKERN_LIB=$(lib_of $name) python bench.py --no-cpu-baseline --no-secondary --ragged --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('ragged %-6s %.3f ms/step  %.0f valid frames/s' % ('$name', d['ms_per_step'], d['value']))
" | tee -a $OUT
done
for name in ring2 ring3; do
  echo "== pytest -k lstm on $name" | tee -a $OUT
  SEPKERN_LIB=$(lib_of $name) python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -k "lstm or configs_match or reference" -x -q 2>&1 | tail -3 | tee -a $OUT
done
