#!/bin/bash
# stream-K kernel (variant 6) beside the chosen kernels (0) and the plain 256 x 256 kernel (5), interleaved
cd "$(dirname "$0")/../.."
S="--shape 12800,7168,1792,0,1 --shape 12800,1792,7168,0,0 --shape 16384,8192,2048,0,1 --shape 8192,8192,8192,0,1 --shape 7168,1792,12800,1,0"
for rep in 1 2; do
  for v in 0 5 6; do
    echo "== variant $v"; timeout -k 10 120 python speech-separation_amd/tools/gemm_bench.py --variant $v $S || exit 1
  done
done
