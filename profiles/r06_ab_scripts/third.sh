#!/bin/bash
# r06, third GPU call: the split-product backward recurrence, software-pipelined (pair p split while pair p - 1's products run)
# against its first form (split, then multiply) and the fp32-MFMA kernel: stand-alone (one layer, us per time step), in the step.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
S=$R/speech-separation_amd/sepkern
cd $R
OUT=$O/r06_bwd_split_pipelined.txt
: > $OUT
echo "== stand-alone, product build (S3 = pipelined)" | tee -a $OUT
timeout -k 10 200 python speech-separation_amd/tools/lstm_bench.py --rounds 5 --fwd "0,1,1,0,0,0,0,0,1" --bwd "0,1,0,0,0,0,31,0,0;0,1,0,0,0,0,31,0,1" 2>&1 | grep -v amdgpu.ids | tee -a $OUT
timeout -k 10 200 python speech-separation_amd/tools/lstm_bench.py --ragged --rounds 5 --fwd "0,1,1,0,0,0,0,0,1" --bwd "0,1,0,0,0,0,31,0,0;0,1,0,0,0,0,31,0,1" 2>&1 | grep -v amdgpu.ids | tee -a $OUT
echo "== stand-alone, S3 in its first form (SK_BWD_S3_PIPE=0)" | tee -a $OUT
SEPKERN_ALLOW_DIAGNOSTIC_LIB=1 SEPKERN_LIB=$S/libsepkern_s3nopipe.so timeout -k 10 200 python speech-separation_amd/tools/lstm_bench.py --rounds 5 --fwd "0,1,1,0,0,0,0,0,1" --bwd "0,1,0,0,0,0,31,0,0;0,1,0,0,0,0,31,0,1" 2>&1 | grep -v amdgpu.ids | tee -a $OUT
line() { python -c "
import json,sys
l=[x for x in sys.stdin.read().strip().splitlines() if x.startswith('{')]
if not l: print('$1: no line'); sys.exit(0)
d=json.loads(l[-1]); k=d['kernels']
print('%-10s %.3f ms/step  loss %.6f  ' % ('$1', d['ms_per_step'], d['config']['mean_loss']) + '  '.join('%s %.2f' % (n, v['ms_per_step']) for n, v in k.items()))
"; }
echo "== in the step" | tee -a $OUT
for i in 1 2 3; do
  SEPKERN_LSTM_BWD_TOP=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-aux --no-power-probe --steps 20 --warmup 3 2>/dev/null | line fp32top | tee -a $OUT
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-aux --no-power-probe --steps 20 --warmup 3 2>/dev/null | line splittop | tee -a $OUT
done
for name in fp32top splittop fp32top splittop; do
  v=1; [ $name = fp32top ] && v=0
  SEPKERN_LSTM_BWD_TOP=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-aux --no-power-probe --ragged --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1])
print('ragged %-9s %.3f ms/step  %.0f valid frames/s  ' % ('$name', d['ms_per_step'], d['value']) + '  '.join('%s %.2f' % (n, v['ms_per_step']) for n, v in d['kernels'].items() if 'lstm' in n))
" | tee -a $OUT
done
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_signed_error.py -m gpu -q -k "lstm or recurrence" > $O/r06c_tests.log 2>&1; echo "pytest rc $?" | tee -a $O/r06c_tests.log; tail -3 $O/r06c_tests.log
