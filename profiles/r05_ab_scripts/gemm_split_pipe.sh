#!/bin/bash
# the software-pipelined form of the 128 x 128 split kernel (-DSK_SPLIT_PIPE=3|4 LDS stages) against the plain form: tests, stand-alone,
# sustained with power, and the training step
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_gemm_split_pipe.txt
T=$R/speech-separation_amd/tools; L=$R/speech-separation_amd/sepkern
: > $O
S="--shape 12800,7168,1792,0,1 --shape 12800,1792,7168,0,0 --shape 7168,1792,12800,1,0,4 --shape 3584,896,12800,1,0,5 --shape 12800,514,1792,0,1"
for v in pipe4 pipe3; do
  SEPKERN_LIB=$L/libsepkern_$v.so python3 -m pytest $R/tests/test_gpu_kernels.py -m gpu -x -q -k "gemm" > $R/gpurun_out/r05_pipe_tests_$v.log 2>&1; echo "tests $v rc=$?" >> $O; tail -1 $R/gpurun_out/r05_pipe_tests_$v.log >> $O
done
for v in default pipe4 pipe3; do
  if [ $v = default ]; then unset SEPKERN_LIB; else export SEPKERN_LIB=$L/libsepkern_$v.so; fi
  echo "== $v (variant 2)" >> $O; python3 $T/gemm_bench.py --variant 2 $S >> $O 2>/dev/null || exit 1
done
for v in default pipe4; do
  if [ $v = default ]; then unset SEPKERN_LIB; else export SEPKERN_LIB=$L/libsepkern_$v.so; fi
  echo "== sustained $v" >> $O; python3 $T/gemm_power.py 4 2>/dev/null | grep -A3 "three-way" | cut -c1-260 >> $O
done
for i in 1 2; do
  for v in default pipe4 pipe3; do
    if [ $v = default ]; then unset SEPKERN_LIB; else export SEPKERN_LIB=$L/libsepkern_$v.so; fi
    python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2> /dev/null |
      python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('%-9s run $i: %.3f ms/step  %.0f frames/s  ' % ('$v', d['ms_per_step'], d['value']) + '  '.join('%s %.2f' % (n, v['ms_per_step']) for n, v in sorted(k.items()) if v['ms_per_step'] > 1.0))" >> $O || exit 1
  done
done
unset SEPKERN_LIB
cat $O
