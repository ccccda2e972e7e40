#!/bin/bash
# One call: upper layers' weight reorders ahead on the side stream (default) vs in front of each projection.
cd "$(dirname "$0")/../.."
run() { timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['mean_loss'])"; }
for rep in 1 2 3; do
  echo "== ahead (default)"; run
  echo "== SEPKERN_PREP_AHEAD=0"; SEPKERN_PREP_AHEAD=0 run
done
