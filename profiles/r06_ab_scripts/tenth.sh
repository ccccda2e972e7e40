#!/bin/bash
# r06, tenth GPU call (run twice: tag r06h, and r06i after the last source edit): the whole GPU suite on the final tree (incl. the census comparison against DESIGN section 4), then the
# round's collection with the corrected trace commands (profiles/collect.sh r06i).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/r06i_gpu_suite.log 2>&1; rc=$?; echo "gpu suite rc $rc"; tail -3 $O/r06i_gpu_suite.log
[ $rc = 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r06i_smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/r06i_smoke.log
bash profiles/collect.sh r06i; echo "collect rc $?"
