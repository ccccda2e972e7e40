#!/bin/bash
# A/B of the bf16 configuration (BASELINE configs[3]) in ONE call on one device: K-major products on the row-major
# copies (r03) vs transposed copies (r02).
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  for km in 1 0; do
    echo "== SEPKERN_BF16_KMAJOR=$km (rep $rep)"
    SEPKERN_BF16_KMAJOR=$km timeout -k 10 200 python bench.py --dtype bf16 --num-spk 3 --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['mean_loss'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})"
  done
done
