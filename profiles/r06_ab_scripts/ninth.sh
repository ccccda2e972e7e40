#!/bin/bash
# r06, ninth GPU call: sk_gemm_pl3 with 32-k stages (64-byte row segments): parity, per-shape timing against variant 9, A/B in the step.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "pl3 or arrive_split" > $O/r06i_tests_quick.log 2>&1; rc=$?; echo "pytest quick rc $rc"; tail -4 $O/r06i_tests_quick.log
[ $rc = 0 ] || exit 1
timeout -k 10 300 python speech-separation_amd/tools/gemm_bench.py --pl3 > $O/r06i_gemm_bench_pl3.txt 2>&1; cat $O/r06i_gemm_bench_pl3.txt
OUT=$O/r06_main_planes_bk32.txt
: > $OUT
line() { python -c "
import json,sys
l=[x for x in sys.stdin.read().strip().splitlines() if x.startswith('{')]
if not l: print('$1: no line'); sys.exit(0)
d=json.loads(l[-1]); k=d['kernels']
print('%-16s %.3f ms/step  %.0f frames/s  loss %.6f  ' % ('$1', d['ms_per_step'], d['value'], d['config']['mean_loss']) + '  '.join('%s %.2f' % (n, v['ms_per_step']) for n, v in k.items()))
"; }
for i in 1 2 3; do
  SEPKERN_MAIN_PLANES=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-aux --no-power-probe --steps 20 --warmup 3 2>/dev/null | line staged | tee -a $OUT
  SEPKERN_MAIN_PLANES=1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-aux --no-power-probe --steps 20 --warmup 3 2>/dev/null | line mainplanes | tee -a $OUT
done
for name in staged mainplanes staged mainplanes; do
  v=1; [ $name = staged ] && v=0
  SEPKERN_MAIN_PLANES=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-aux --no-power-probe --ragged --steps 20 --warmup 3 2>/dev/null | line ragged_$name | tee -a $OUT
done
