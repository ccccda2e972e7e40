#!/bin/bash
# r06, eighth GPU call: the main-stream products on planes (sk_gemm_pl3): parity, then A/B in the step (SEPKERN_MAIN_PLANES=0: the
# 256 x 128 split-while-staging kernel on fp32 operands); ragged; RSH.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -m gpu -x -q -k "pl3 or arrive_split or configs_match or reference or train or fp32_step_32x400_matches or rsh" > $O/r06h_tests_quick.log 2>&1; rc=$?; echo "pytest quick rc $rc"; tail -4 $O/r06h_tests_quick.log
[ $rc = 0 ] || exit 1
OUT=$O/r06_main_planes.txt
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
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-aux --no-power-probe --steps 20 --warmup 3 2>/dev/null | line mainplanes | tee -a $OUT
done
for name in staged mainplanes staged mainplanes; do
  v=1; [ $name = staged ] && v=0
  SEPKERN_MAIN_PLANES=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-aux --no-power-probe --ragged --steps 20 --warmup 3 2>/dev/null | line ragged_$name | tee -a $OUT
done
for name in staged mainplanes; do
  v=1; [ $name = staged ] && v=0
  SEPKERN_MAIN_PLANES=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-aux --no-power-probe --arch rsh --hidden 600 --layers 2 --num-spk 4 --steps 10 --warmup 2 2>/dev/null | line rsh_$name | tee -a $OUT
  SEPKERN_MAIN_PLANES=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-aux --no-power-probe --hidden 600 --layers 2 --batch 100 --steps 10 --warmup 2 2>/dev/null | line b100_$name | tee -a $OUT
done
python speech-separation_amd/tools/gemm_bench.py > $O/r06h_gemm_bench.txt 2>&1 || true
