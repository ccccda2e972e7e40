#!/bin/bash
# One call, one device: the fp32 step with the side-stream GEMM kernel choices, and the bf16 step with / without
# K-major products.
cd "$(dirname "$0")/../.."
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['mean_loss'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})"; }
for rep in 1 2; do
  for v in "0,1" "0,0" "0,4"; do echo "== fp32 SEPKERN_GEMM_VARIANTS=$v"; SEPKERN_GEMM_VARIANTS=$v run; done
  for km in 1 0; do echo "== bf16 3-spk SEPKERN_BF16_KMAJOR=$km"; SEPKERN_BF16_KMAJOR=$km run --dtype bf16 --num-spk 3; done
done
cd speech-separation_amd && timeout -k 10 300 python tools/gemm_bench.py --km 2>&1 | grep -v amdgpu
