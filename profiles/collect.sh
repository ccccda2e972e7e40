#!/bin/bash
# Collects what profiles/ holds for one round tag (run on the GPU box through gpurun):
#   the two separate PMC passes (FETCH_SIZE, WRITE_SIZE) for f32 and bf16 FIRST -- profiles/pmc_traffic.py then makes
#   pmc_traffic.json for these sources on the box, so that the bench lines collected afterwards carry `roofline.traffic` --
#   then the bench lines and the rocprofv3 kernel stats.  profiles/install.sh <tag> installs the result here afterwards.
# usage: profiles/collect.sh <tag>
set -o pipefail
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for dt in f32 bf16; do
  for c in FETCH_SIZE WRITE_SIZE; do
    spk=2; [ $dt = bf16 ] && spk=3      # BASELINE configs[3]: the bf16 configuration has 3 speakers
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/prof_${tag}_${dt}_$c -- python3 $R/bench.py --dtype $dt --num-spk $spk --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-secondary > /dev/null 2> $O/prof_${tag}_${dt}_$c.err || exit 3
  done
done
python3 $R/profiles/pmc_traffic.py $tag > $O/pmc_traffic_$tag.txt || exit 6
cp $R/profiles/pmc_traffic.json $O/pmc_traffic_$tag.json
python3 $R/bench.py > $O/bench_$tag.json 2> $O/bench_$tag.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-aux --no-power-probe > $O/prof_${tag}_trace.json 2> $O/prof_${tag}_trace.err || exit 2
python3 $R/bench.py --ragged --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_${tag}_ragged.json 2> $O/bench_${tag}_ragged.err || exit 7
python3 $R/bench.py --ragged --padded-rows --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${tag}_ragged_padded_rows.json 2> /dev/null || exit 8
SEPKERN_LSTM_FWD=0,1,1,0,0,0,0,0 SEPKERN_GEMM_VARIANTS=8,1 python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench_${tag}_plain_fp32_mfma.json 2> /dev/null || exit 9
python3 $R/bench.py --dtype bf16 --num-spk 3 --steps 20 --no-cpu-baseline > $O/bench_${tag}_bf16.json 2> $O/bench_${tag}_bf16.err || exit 4
python3 $R/bench.py --arch rsh --hidden 600 --layers 2 --num-spk 4 --steps 20 --no-cpu-baseline > $O/bench_${tag}_rsh_4spk.json 2> /dev/null || exit 10
python3 $R/bench.py --ragged --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_${tag}_bf16_ragged.json 2> /dev/null || exit 11
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_bf16_trace -- python3 $R/bench.py --dtype bf16 --num-spk 3 --steps 5 --warmup 2 --no-cpu-baseline --no-aux > $O/prof_${tag}_bf16_trace.json 2> $O/prof_${tag}_bf16_trace.err || exit 5
echo collected $tag
