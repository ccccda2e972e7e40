#!/bin/bash
# A/B of the read-ahead in the fp32 backward recurrence's product (SK_BWD_RING=1: chunk j + 1's fragment read issued before chunk
# j's four MFMAs): training step, three alternations; ragged; the recurrence parity tests on the build.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_bwd_ring.txt
: > $OUT
cd $R
for i in 1 2 3; do
  for name in ring0 bring; do
    lib=$R/speech-separation_amd/sepkern/libsepkern.so; [ $name = bring ] && lib=$R/speech-separation_amd/sepkern/libsepkern_bring.so
    SEPKERN_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernels']
print('%-6s %.3f ms/step  loss %.6f  ' % ('$name', d['ms_per_step'], d['config']['mean_loss']) + '  '.join('%s %.2f' % (n, v['ms_per_step']) for n, v in k.items()))
" | tee -a $OUT
  done
done
for name in ring0 bring; do
  lib=$R/speech-separation_amd/sepkern/libsepkern.so; [ $name = bring ] && lib=$R/speech-separation_amd/sepkern/libsepkern_bring.so
  SEPKERN_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --ragged --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print('ragged %-6s %.3f ms/step  %.0f valid frames/s' % ('$name', d['ms_per_step'], d['value']))
" | tee -a $OUT
done
echo "== pytest -k lstm on bring" | tee -a $OUT
SEPKERN_LIB=$R/speech-separation_amd/sepkern/libsepkern_bring.so timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -k "lstm or configs_match or reference" -x -q 2>&1 | tail -3 | tee -a $OUT
