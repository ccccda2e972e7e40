#!/bin/bash
# One call: upper layers' weight reorders ahead on the side stream (default) vs in front of each projection.
cd "$(dirname "$0")/../.."
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['mean_loss'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})"; }
for rep in 1 2 3; do
  echo "== default"; run
  echo "== SEPKERN_GEMM_STREAMK=0"; SEPKERN_GEMM_STREAMK=0 run
done
