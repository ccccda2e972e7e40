#!/bin/bash
# PMC passes over tools/stream_bench.py (STFT / mask-iSTFT / PIT-MSE at corpus scale): what binds the streaming kernels.
# Separate --pmc passes (no trace domains beside --kernel-trace), summarised by profiles/stream_pmc.py.
# usage (on the GPU box, through gpurun): profiles/stream_pmc.sh <tag>
set -o pipefail
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/prof_${tag}_stream_$i -- python3 $R/speech-separation_amd/tools/stream_bench.py > $O/prof_${tag}_stream_$i.log 2> $O/prof_${tag}_stream_$i.err || exit $i
done
python3 $R/speech-separation_amd/tools/stream_bench.py > $O/stream_bench_$tag.txt 2>/dev/null
echo collected $tag
