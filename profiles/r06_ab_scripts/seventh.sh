#!/bin/bash
# r06, seventh GPU call: the census of the final tree, then the round's collection (profiles/collect.sh r06g).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 300 python tests/test_gpu_census.py > $O/r06g_census_stdout.md 2> $O/r06g_census.err || { echo "census failed"; tail -5 $O/r06g_census.err; }
bash profiles/collect.sh r06g; echo "collect rc $?"
