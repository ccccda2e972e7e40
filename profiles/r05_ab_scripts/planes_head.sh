#!/bin/bash
# planes kernel: N VALU instructions of the split stated BEFORE the step's first MFMA (in the shadow of its first fragment reads),
# then one MFMA + S VALU: hNsS builds against the shipped (head 0, 4 VALU per MFMA).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_planes_head.txt
: > $OUT
cd $R/speech-separation_amd/tools
for rep in 1 2; do
for name in default h16s3 h8s4 h16s4 h24s3; do
  lib=$R/speech-separation_amd/sepkern/libsepkern.so; [ $name != default ] && lib=$R/speech-separation_amd/sepkern/libsepkern_$name.so
  echo "== $name" | tee -a $OUT
  SEPKERN_LIB=$lib timeout -k 10 120 python gemm_bench.py --variant 9 --shape 12800,1792,7168,0,0 --shape 12800,7168,1792,0,1 --shape 7168,1792,12800,1,0 --shape 8192,8192,8192,0,1 2>&1 | grep custom | tee -a $OUT
done
done
