#!/bin/bash
# The PMC half of profiles/collect.sh alone (separate FETCH_SIZE / WRITE_SIZE passes, f32 and bf16) + pmc_traffic.py: refreshes
# profiles/pmc_traffic.json for the current kernel sources after a source change that leaves the timings of a collection valid.
# usage (on the GPU box): profiles/collect_pmc.sh <tag>; then copy gpurun_out/pmc_traffic_<tag>.json to profiles/pmc_traffic.json
set -o pipefail
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for dt in f32 bf16; do
  for c in FETCH_SIZE WRITE_SIZE; do
    spk=2; [ $dt = bf16 ] && spk=3
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/prof_${tag}_${dt}_$c -- python3 $R/bench.py --dtype $dt --num-spk $spk --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-secondary > /dev/null 2> $O/prof_${tag}_${dt}_$c.err || exit 3
  done
done
python3 $R/profiles/pmc_traffic.py $tag > $O/pmc_traffic_$tag.txt || exit 6
cp $R/profiles/pmc_traffic.json $O/pmc_traffic_$tag.json
echo collected pmc $tag
