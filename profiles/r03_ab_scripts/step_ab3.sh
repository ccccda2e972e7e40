#!/bin/bash
# One call: the exact three-way bf16 split kernel (fp32 products on the bf16 matrix pipe) for the hosted products only /
# for every product, on the r03 kernels.
cd "$(dirname "$0")/../.."
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['mean_loss'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})"; }
for rep in 1 2; do
  for v in "0,1" "0,2" "2,2"; do echo "== fp32 SEPKERN_GEMM_VARIANTS=$v"; SEPKERN_GEMM_VARIANTS=$v run; done
done
