#!/bin/bash
# A/B on ONE device in ONE call (devices differ by up to 12 %, cdna_hip_programming.md 5.4 rule 24):
#   usage: ab_bench.sh <rounds> <name>=<dir>[,ENV=val...] ...     e.g.  ab_bench.sh 2 A=_ab/base C=. D=.,SEPKERN_OVERLAP=0
cd $GRAFT_REPO_ROOT
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for spec in "$@"; do
    name=${spec%%=*}; rest=${spec#*=}
    dir=${rest%%,*}; envs=""
    if [ "$rest" != "$dir" ]; then envs=$(echo "${rest#*,}" | tr ',' ' ' | tr '+' ','); fi
    (cd $dir && env $envs timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $BENCH_ARGS 2>/dev/null) > gpurun_out/ab_${name}${r}.json
    python3 -c "
import json
d=json.load(open('gpurun_out/ab_${name}${r}.json'))
print('${name}${r}', d['value'], d['ms_per_step'], {n:(v['launches_per_step'], v['ms_per_step']) for n,v in d['kernels'].items()})"
  done
done
