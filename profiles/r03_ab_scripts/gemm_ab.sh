#!/bin/bash
# fp32 GEMM kernels on the step's large products, variants interleaved in one call: 0 = the shipped choice, 4 = 256 x 128
# tiles, 5 = 256 x 256 tiles (r03), 1 = register-staged 128 x 128.
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for v in 0 4 5 1; do
    echo "== variant $v"
    timeout -k 10 120 python tools/gemm_bench.py --variant $v --shape 12800,7168,1792,0,1 --shape 12800,1792,7168,0,0 --shape 16384,8192,2048,0,1 --shape 8192,8192,8192,0,0 2>&1 | grep -v amdgpu
  done
done
