#!/bin/bash
# Runs the GPU parity tests in separate processes per group (an aborting kernel must not hide the others).
# Usage: tests/run_gpu_groups.sh <tag> [groups...]; logs go to gpurun_out/<tag>_<group>.log
tag=$1; shift
mkdir -p gpurun_out
groups=${@:-"stft istft pit bn_or_colsum_or_clip lstm model rsh edges"}
for g in $groups; do
  if [ "$g" = "model" ]; then sel="tests/test_gpu_model.py"; k=""; elif [ "$g" = "rsh" ]; then sel="tests/test_gpu_rsh.py"; k=""; elif [ "$g" = "edges" ]; then sel="tests/test_gpu_edges.py"; k=""; else sel="tests/test_gpu_kernels.py"; k="-k ${g//_or_/ or }"; fi
  timeout -k 10 400 python -m pytest $sel -m gpu -q --timeout 180 -p no:cacheprovider --tb=short $k > gpurun_out/${tag}_${g}.log 2>&1
  rc=$?
  echo "== $g rc=$rc: $(tail -1 gpurun_out/${tag}_${g}.log)"
  if [ $rc -ge 124 ]; then echo "stopping after timeout/kill in $g"; exit 1; fi
done
exit 0
