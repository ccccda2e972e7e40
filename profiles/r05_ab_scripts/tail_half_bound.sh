#!/bin/bash
# Upper bound of re-partitioning the TAIL of a ragged batch, one gpurun call, three alternations:
#   A  the shipped library
#   B  libsepkern_tailhalf.so (make -C speech-separation_amd/csrc variant NAME=tailhalf DEFS=-DSK_TAIL_HALF): TIMING ONLY, wrong
#      numerics -- once the short batch group's streams have left, the long group's recurrences issue half their MFMAs
# on `bench.py --ragged` (fp32) and `--ragged --dtype bf16`.  Output: gpurun_out/r05_tail_half_bound.txt
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_tail_half_bound.txt
L=$R/speech-separation_amd/sepkern/libsepkern_tailhalf.so
: > $O
for dt in f32 bf16; do
  for i in 1 2 3; do
    for v in A B; do
      if [ $v = B ]; then export SEPKERN_LIB=$L; else unset SEPKERN_LIB; fi
      python3 $R/bench.py --ragged --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline 2> /dev/null |
        python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); by=d['roofline']['by_kernel']
print('$dt $v run $i: %.3f ms/step  %.0f valid frames/s   fwd rec %.3f ms (%.2f us/step)  bwd rec %.3f ms (%.2f us/step)' % (d['ms_per_step'], d['value'], by['lstm_fwd_kernel']['ms_per_step'], by['lstm_fwd_kernel']['us_per_time_step'], by['lstm_bwd_kernel']['ms_per_step'], by['lstm_bwd_kernel']['us_per_time_step']))" >> $O || exit 1
    done
  done
done
unset SEPKERN_LIB
cat $O
