#!/bin/bash
# A/B of the fragment-read rings in the bf16 recurrences (SK_FWD_RING_BF=6: forward, reads in flight ahead of their products;
# SK_BWD_BATCH_BF=1: backward, a sub-block's reads issued together): bf16 3-speaker configuration, three alternations; the bf16
# parity tests on the ring build.  ("rings" also has SK_FWD_RING=3 for the fp32 split forward recurrence.)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_bf16_rings.txt
: > $OUT
lib_of() { if [ $1 = ring0 ]; then echo $R/speech-separation_amd/sepkern/libsepkern.so; else echo $R/speech-separation_amd/sepkern/libsepkern_$1.so; fi; }
cd $R
for i in 1 2 3; do
  for name in ring0 rings; do
    SEPKERN_LIB=$(lib_of $name) python bench.py --no-cpu-baseline --no-secondary --dtype bf16 --num-spk 3 --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
k=d['kernels']
print('%-6s %.3f ms/step  loss %.6f  ' % ('$name', d['ms_per_step'], d['config']['mean_loss']) + '  '.join('%s %.2f' % (n, v['ms_per_step']) for n, v in k.items()))
" | tee -a $OUT
  done
done
for name in ring0 rings; do
  SEPKERN_LIB=$(lib_of $name) python bench.py --no-cpu-baseline --no-secondary --dtype bf16 --ragged --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('ragged bf16 %-6s %.3f ms/step  %.0f valid frames/s' % ('$name', d['ms_per_step'], d['value']))
" | tee -a $OUT
done
echo "== pytest -k 'bf16 or lstm' on rings" | tee -a $OUT
SEPKERN_LIB=$(lib_of rings) python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -k "bf16 or lstm" -x -q 2>&1 | tail -3 | tee -a $OUT
