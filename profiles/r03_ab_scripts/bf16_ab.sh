#!/bin/bash
# bf16 3-speaker training step, A/B of environment settings (first = reference), three alternations
cd "$(dirname "$0")/../.."
run() { env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --dtype bf16 --num-spk 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['mean_loss'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})"; }
for rep in 1 2 3; do
  for cfg in "$@"; do
    echo "== $cfg"; run $cfg
  done
done
