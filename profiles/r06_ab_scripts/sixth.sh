#!/bin/bash
# r06, sixth GPU call: the census; the exclusive / co-resident policy of the backward recurrences checked on both workloads;
# the whole GPU suite.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 300 python tests/test_gpu_census.py > $O/r06f_census_stdout.md 2> $O/r06f_census.err || { echo "census failed"; tail -5 $O/r06f_census.err; }
OUT=$O/r06_bwd_exclusive_policy.txt
: > $OUT
line() { python -c "
import json,sys
l=[x for x in sys.stdin.read().strip().splitlines() if x.startswith('{')]
if not l: print('$1: no line'); sys.exit(0)
d=json.loads(l[-1]); k=d['kernels']
print('%-22s %.3f ms/step  %.0f frames/s  ' % ('$1', d['ms_per_step'], d['value']) + '  '.join('%s %.2f' % (n, v['ms_per_step']) for n, v in k.items()))
"; }
for i in 1 2; do
  SEPKERN_WGRAD_PLANES=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-aux --no-power-probe --steps 20 --warmup 3 2>/dev/null | line uniform_r05_fp32ops | tee -a $OUT
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-aux --no-power-probe --steps 20 --warmup 3 2>/dev/null | line uniform_default | tee -a $OUT
  SEPKERN_BWD_EXCLUSIVE=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-aux --no-power-probe --steps 20 --warmup 3 2>/dev/null | line uniform_exclusive | tee -a $OUT
  SEPKERN_WGRAD_PLANES=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-aux --no-power-probe --ragged --steps 20 --warmup 3 2>/dev/null | line ragged_r05_fp32ops | tee -a $OUT
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-aux --no-power-probe --ragged --steps 20 --warmup 3 2>/dev/null | line ragged_default | tee -a $OUT
  SEPKERN_BWD_EXCLUSIVE=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-aux --no-power-probe --ragged --steps 20 --warmup 3 2>/dev/null | line ragged_coresident | tee -a $OUT
done
timeout -k 10 1000 python -m pytest tests -m gpu -q --deselect tests/test_gpu_census.py::test_design_md_carries_the_generated_census > $O/r06f_tests.log 2>&1; echo "pytest rc $?" | tee -a $O/r06f_tests.log; tail -8 $O/r06f_tests.log
