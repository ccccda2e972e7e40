#!/bin/bash
# r06, fifth GPU call: weight gradients on planes, now CO-RESIDENT with the backward recurrence (its G = 1 instantiation keeps
# 113 instead of 127 KB of LDS: the 48 KB of gemm_f32_kernel_pl3 fit beside it) -- A/B in the step; PIT pair kernel (fixed-bin
# sweeps); then the whole GPU suite with durations.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "lstm or pit or arrive_split or configs_match or reference or xcd_local" > $O/r06e_tests_quick.log 2>&1; rc=$?; echo "pytest quick rc $rc"; tail -3 $O/r06e_tests_quick.log
[ $rc = 0 ] || exit 1
OUT=$O/r06_wgrad_planes_coresident.txt
: > $OUT
line() { python -c "
import json,sys
l=[x for x in sys.stdin.read().strip().splitlines() if x.startswith('{')]
if not l: print('$1: no line'); sys.exit(0)
d=json.loads(l[-1]); k=d['kernels']
print('%-10s %.3f ms/step  loss %.6f  ' % ('$1', d['ms_per_step'], d['config']['mean_loss']) + '  '.join('%s %.2f' % (n, v['ms_per_step']) for n, v in k.items()))
"; }
for i in 1 2 3; do
  SEPKERN_WGRAD_PLANES=0 timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-aux --no-power-probe --steps 20 --warmup 3 2>/dev/null | line fp32ops | tee -a $OUT
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-aux --no-power-probe --steps 20 --warmup 3 2>/dev/null | line planes | tee -a $OUT
done
for name in fp32ops planes fp32ops planes; do
  v=1; [ $name = fp32ops ] && v=0
  SEPKERN_WGRAD_PLANES=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-aux --no-power-probe --ragged --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1])
print('ragged %-9s %.3f ms/step  %.0f valid frames/s  ' % ('$name', d['ms_per_step'], d['value']) + '  '.join('%s %.2f' % (n, v['ms_per_step']) for n, v in d['kernels'].items()))
" | tee -a $OUT
done
OUT3=$O/r06_bf16_xl8_bwd.txt
: > $OUT3
timeout -k 10 200 python speech-separation_amd/tools/lstm_bench.py --bf16 --rounds 5 --fwd "0,1,1,0,0,1,0,0,0,1" --bwd "0,1,0,0,0,0,31,0,0,0;0,1,0,0,0,0,31,0,0,1;0,1,0,0,0,0,4,0,0,1;0,1,0,0,0,0,8,0,0,1" 2>&1 | grep -v amdgpu.ids | tee -a $OUT3
timeout -k 10 200 python speech-separation_amd/tools/lstm_bench.py --bf16 --ragged --rounds 5 --fwd "0,1,1,0,0,1,0,0,0,1" --bwd "0,1,0,0,0,0,31,0,0,0;0,1,0,0,0,0,31,0,0,1" 2>&1 | grep -v amdgpu.ids | tee -a $OUT3
for i in 1 2; do
  SEPKERN_LSTM_BWD=0,1,0,0,0,31,0,0,0 timeout -k 10 200 python bench.py --dtype bf16 --num-spk 3 --no-cpu-baseline --no-secondary --no-aux --no-power-probe --steps 20 --warmup 3 2>/dev/null | line bf16_fwdxl8 | tee -a $OUT3
  timeout -k 10 200 python bench.py --dtype bf16 --num-spk 3 --no-cpu-baseline --no-secondary --no-aux --no-power-probe --steps 20 --warmup 3 2>/dev/null | line bf16_bothxl8 | tee -a $OUT3
done
timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --no-power-probe --steps 10 > $O/r06e_bench.json 2> $O/r06e_bench.err; echo "bench rc $?"; python -c "
import json; d=json.load(open('$O/r06e_bench.json')); print(d['ms_per_step'], {k:(v['us_per_launch'], v['frac_of_hbm_peak']) for k,v in d['aux'].items() if k!='note'})"
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=15 --deselect tests/test_gpu_census.py::test_design_md_carries_the_generated_census > $O/r06e_tests.log 2>&1; echo "pytest rc $?" | tee -a $O/r06e_tests.log; tail -30 $O/r06e_tests.log
