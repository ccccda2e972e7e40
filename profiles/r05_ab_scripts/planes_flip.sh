#!/bin/bash
# A/B of the sign-alternating accumulation in the N/N instantiation of gemm_f32_kernel_planes (SK_PLANES_FLIP = 0 / 32 / 8 K steps
# per phase): error + signed bias against fp64, speed on the step's three large shapes (only N/N changes), and the full-size
# step's gradient error against the oracle (fp32, and float64 for the shipped build).
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_planes_flip2.txt
: > $OUT
lib_of() { if [ $1 = flip32 ]; then echo $R/speech-separation_amd/sepkern/libsepkern.so; else echo $R/speech-separation_amd/sepkern/libsepkern_$1.so; fi; }
cd $R/speech-separation_amd/tools
for name in noflip flip32 flip8 noflip flip32; do
  echo "== $name" | tee -a $OUT
  SEPKERN_LIB=$(lib_of $name) python gemm_bench.py --variant 9 --shape 12800,1792,7168,0,0 --shape 12800,7168,1792,0,1 --shape 7168,1792,12800,1,0 2>&1 | grep custom | tee -a $OUT
done
for name in noflip flip32 flip8; do
  echo "== $name" | tee -a $OUT
  SEPKERN_LIB=$(lib_of $name) python gemm_bias_check.py 8 9 2>&1 | grep "K=" | tee -a $OUT
done
cd $R
for name in noflip flip32; do
  echo "== grad_check 3x896 32x400 (fp32 oracle) $name" | tee -a $OUT
  SEPKERN_LIB=$(lib_of $name) python tests/grad_check.py 896 3 2 32 400 2>&1 | grep "rel err\|loss" | tee -a $OUT
done
python -m pytest tests/test_gpu_fullsize.py::test_fp32_step_32x400_matches_oracle tests/test_gpu_kernels.py -k "gemm or fp32_step" -x -q -s 2>&1 | grep "fullsize fp32\|mean signed\|passed\|failed\|Error" | tee -a $OUT
echo "== grad_check 3x896 32x400 (float64 oracle) flip32" | tee -a $OUT
GRAD_CHECK_F64=1 python tests/grad_check.py 896 3 2 32 400 2>&1 | grep "rel err\|loss" | tee -a $OUT
