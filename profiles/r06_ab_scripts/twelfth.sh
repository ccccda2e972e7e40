#!/bin/bash
# r06, twelfth GPU call: where the GPU suite's 460-480 s go (durations of the slowest tests)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=40 > $O/r06i_durations.log 2>&1; echo "rc $?"; tail -60 $O/r06i_durations.log
