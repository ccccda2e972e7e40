#!/bin/bash
# r06, thirteenth GPU call: shader clock / package power while the phases of the fp32 step run on their own (tools/step_clocks.py)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 300 python speech-separation_amd/tools/step_clocks.py --seconds 2.5 > $O/r06_step_clocks.txt 2>&1; echo "rc $?"; cat $O/r06_step_clocks.txt
