#!/bin/bash
# r06, second GPU call: the split-product backward recurrence for the top layer, built -- A/B in the step (three alternations);
# signed-error table on the final sign-phase rule; the whole GPU suite; the census; the default bench line (no CPU baseline).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
OUT=$O/r06_bwd_split_top_layer_built.txt
: > $OUT
line() { python -c "
import json,sys
l=[x for x in sys.stdin.read().strip().splitlines() if x.startswith('{')]
if not l: print('$1: no line'); sys.exit(0)
d=json.loads(l[-1]); k=d['kernels']
print('%-10s %.3f ms/step  loss %.6f  ' % ('$1', d['ms_per_step'], d['config']['mean_loss']) + '  '.join('%s %.2f' % (n, v['ms_per_step']) for n, v in k.items()))
"; }
for i in 1 2 3; do
  SEPKERN_LSTM_BWD_TOP=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-aux --no-power-probe --steps 20 --warmup 3 2>/dev/null | line fp32top | tee -a $OUT
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-aux --no-power-probe --steps 20 --warmup 3 2>/dev/null | line splittop | tee -a $OUT
done
for name in fp32top splittop; do
  v=1; [ $name = fp32top ] && v=0
  SEPKERN_LSTM_BWD_TOP=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-aux --no-power-probe --ragged --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1])
print('ragged %-9s %.3f ms/step  %.0f valid frames/s' % ('$name', d['ms_per_step'], d['value']))
" | tee -a $OUT
done
timeout -k 10 500 python speech-separation_amd/tools/signed_error.py > $O/r06b_signed_error.txt 2>&1 || { echo "signed_error failed"; tail -5 $O/r06b_signed_error.txt; }
timeout -k 10 200 python tests/test_gpu_census.py > $O/r06b_census_stdout.md 2> $O/r06b_census.err || { echo "census failed"; tail -5 $O/r06b_census.err; }
timeout -k 10 1000 python -m pytest tests -m gpu -q --deselect tests/test_gpu_census.py::test_design_md_carries_the_generated_census > $O/r06b_tests.log 2>&1; echo "pytest rc $?" | tee -a $O/r06b_tests.log; tail -15 $O/r06b_tests.log
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/r06b_bench.json 2> $O/r06b_bench.err; echo "bench rc $?"; tail -c 300 $O/r06b_bench.json
