#!/bin/bash
# TIMING-ONLY upper bound (wrong numerics): the split kernels with the pieces for free (-DSK_SPLIT_FREE: no VALU split work; same DMA, same fragment
# reads, same MFMAs) -- what a kernel that finds the three bf16 planes ready in LDS could reach at best.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_gemm_split_free.txt
T=$R/speech-separation_amd/tools; L=$R/speech-separation_amd/sepkern
: > $O
S="--shape 12800,7168,1792,0,1 --shape 12800,1792,7168,0,0 --shape 7168,1792,12800,1,0,4"
for v in default free; do
  if [ $v = default ]; then unset SEPKERN_LIB; else export SEPKERN_LIB=$L/libsepkern_$v.so; fi
  for var in 2 7; do echo "== $v variant $var" >> $O; python3 $T/gemm_bench.py --variant $var $S >> $O 2>/dev/null || exit 1; done
  echo "== sustained $v" >> $O; python3 $T/gemm_power.py 4 2>/dev/null | grep -A4 "three-way" | cut -c1-90,150-400 >> $O
done
for i in 1 2; do
  for v in default free; do
    if [ $v = default ]; then unset SEPKERN_LIB; else export SEPKERN_LIB=$L/libsepkern_$v.so; fi
    python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2> /dev/null |
      python3 -c "
import json,sys
l=[x for x in sys.stdin.read().strip().splitlines() if x.startswith('{')]
if not l: print('$v run $i: no line (loss not finite with wrong numerics?)'); sys.exit(0)
d=json.loads(l[-1]); k=d['kernels']
print('%-9s run $i: %.3f ms/step  %.0f frames/s  ' % ('$v', d['ms_per_step'], d['value']) + '  '.join('%s %.2f' % (n, v['ms_per_step']) for n, v in sorted(k.items()) if v['ms_per_step'] > 1.0))" >> $O
  done
done
unset SEPKERN_LIB
cat $O
