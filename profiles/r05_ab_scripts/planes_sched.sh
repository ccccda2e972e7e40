#!/bin/bash
# A/B of the stated MFMA/VALU interleave (SK_PLANES_SCHED = VALU instructions behind every MFMA) in gemm_f32_kernel_planes.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_planes_sched.txt
: > $OUT
cd $R/speech-separation_amd/tools/micro
timeout -k 10 120 bin/xcd_local_handoff 2000 > $R/gpurun_out/r05_xcd_local_handoff.txt 2>&1
grep "^F\|^B" $R/gpurun_out/r05_xcd_local_handoff.txt | head -4
lib_of() { if [ $1 = default ]; then echo $R/speech-separation_amd/sepkern/libsepkern.so; else echo $R/speech-separation_amd/sepkern/libsepkern_$1.so; fi; }
cd $R/speech-separation_amd/tools
for rep in 1 2; do
for name in noflip default sched2 sched3 sched4; do
  echo "== $name" | tee -a $OUT
  SEPKERN_LIB=$(lib_of $name) python gemm_bench.py --variant 9 --shape 12800,1792,7168,0,0 --shape 12800,7168,1792,0,1 --shape 7168,1792,12800,1,0 --shape 12800,7168,272,0,1 --shape 8192,8192,8192,0,1 2>&1 | grep custom | tee -a $OUT
done
done
