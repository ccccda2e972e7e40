#!/bin/bash
# (the map-3 form this script drives was removed after the run: see profiles/r05_xcc_census_map.txt)
# Block -> stream map 3 (membership by HW_REG_XCC_ID census) against the shipped map 1 (block b assumed on XCD b % 8) in the
# training step: three alternations fp32, two bf16, one ragged; bitwise-variant test first.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_xcc_map.txt
: > $OUT
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -k "geometry_and_protocol" -x -q 2>&1 | tail -4 | tee -a $OUT
grep -q "failed\|error" $OUT && exit 1
run() {  # name, fwd spec, bwd spec, extra bench args
  SEPKERN_LSTM_FWD=$2 SEPKERN_LSTM_BWD=$3 timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 $4 2>/dev/null | python -c "
import json,sys
l=[x for x in sys.stdin.read().strip().splitlines() if x.startswith('{')]
if not l: print('$1: no line'); sys.exit(0)
d=json.loads(l[-1]); k=d['kernels']
print('%-22s %.3f ms/step  loss %.6f  ' % ('$1', d['ms_per_step'], d['config']['mean_loss']) + '  '.join('%s %.2f' % (n, v['ms_per_step']) for n, v in k.items()))
" | tee -a $OUT
}
for i in 1 2 3; do
  run "f32 map1"      0,1,1,0,0,0,0,1 0,1,0,0,0,31,0,0 ""
  run "f32 map3 both" 0,3,1,0,0,0,0,1 0,3,0,0,0,31,0,0 ""
  run "f32 map3 bwd"  0,1,1,0,0,0,0,1 0,3,0,0,0,31,0,0 ""
done
for i in 1 2; do
  run "bf16 map1"      0,1,1,0,1,0,0,0 0,1,0,0,0,31,0,0 "--dtype bf16 --num-spk 3"
  run "bf16 map3 both" 0,3,1,0,1,0,0,0 0,3,0,0,0,31,0,0 "--dtype bf16 --num-spk 3"
done
run "ragged map1"      0,1,1,0,0,0,0,1 0,1,0,0,0,31,0,0 "--ragged"
run "ragged map3 both" 0,3,1,0,0,0,0,1 0,3,0,0,0,31,0,0 "--ragged"
