#!/bin/bash
# (needs profiles/r05_bf16_xcd_local_streams.patch applied: the XL form is not in the tree)
# bf16 forward recurrence with XCD-local streams (mode bit 30, SEPKERN_LSTM_FWD 9th field): parity test, then the bf16 3-speaker
# step with the shipped form and with XL at several hold-backs of the first poll (6th field, x 0.1 us; 0 = library's 0.8, 31 = none).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_bf16_xl.txt
: > $OUT
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -k "xcd_local" -x -q 2>&1 | tail -15 | tee -a $OUT
grep -q "passed" $OUT || exit 1
grep -q "failed" $OUT && exit 1
for i in 1 2; do
  for spec in default 0,1,1,0,1,0,0,0,1 0,1,1,0,1,4,0,0,1 0,1,1,0,1,2,0,0,1 0,1,1,0,1,31,0,0,1; do
    if [ $spec = default ]; then unset SEPKERN_LSTM_FWD; else export SEPKERN_LSTM_FWD=$spec; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --dtype bf16 --num-spk 3 --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
l=[x for x in sys.stdin.read().strip().splitlines() if x.startswith('{')]
if not l: print('$spec: no line'); sys.exit(0)
d=json.loads(l[-1]); k=d['kernels']
print('%-20s %.3f ms/step  loss %.6f  ' % ('$spec', d['ms_per_step'], d['config']['mean_loss']) + '  '.join('%s %.2f' % (n, v['ms_per_step']) for n, v in k.items()))
" | tee -a $OUT
  done
done
unset SEPKERN_LSTM_FWD
