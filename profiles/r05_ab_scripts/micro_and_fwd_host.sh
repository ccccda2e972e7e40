#!/bin/bash
# (1) tools/micro/xcd_local_handoff: the forward hand-off chain alone, shipped geometry vs one-XCD streams through the L2
# (2) profiles/r05_fwd_host_bound.sh: timing-only upper bound of hosting projections beside the forward recurrences
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R/speech-separation_amd/tools/micro
timeout -k 10 120 bin/xcd_local_handoff 2000 > $R/gpurun_out/r05_xcd_local_handoff.txt 2>&1
echo "micro rc $?"
cat $R/gpurun_out/r05_xcd_local_handoff.txt
cd $R && bash profiles/r05_ab_scripts/fwd_host_bound.sh
