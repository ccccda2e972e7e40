#!/bin/bash
# stream-K kernel (variant 6): where the time of the input projection goes (tile count, K, operand form)
cd "$(dirname "$0")/../.."
S="--shape 12800,7168,1792,0,1 --shape 12800,7168,1792,0,0 --shape 11776,7168,1792,0,1 --shape 13056,7168,1792,0,1 --shape 12800,7168,3584,0,1 --shape 16384,8192,1792,0,1 --shape 16384,8192,1792,0,0"
for v in 5 6; do
  echo "== variant $v"; timeout -k 10 120 python speech-separation_amd/tools/gemm_bench.py --variant $v $S 2>/dev/null || exit 1
done
