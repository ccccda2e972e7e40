#!/bin/bash
# loss of EVERY 5th step of 200 optimisation steps on one batch, fp32, under the arithmetic variants: is a bump in the curve the arithmetic or
# the trajectory?  Two h0/c0 seeds each.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_train_curve_variants.txt
T=$R/speech-separation_amd/tools/train_curve.py
: > $O
for seed in 1234 99; do
  for e in "default" "SEPKERN_GEMM_PLANES=0" "SEPKERN_GEMM_SPLIT=0" "SEPKERN_GEMM_SPLIT=0 SEPKERN_LSTM_FWD=0,1,1,0,0,0,0,0" "SEPKERN_LSTM_FWD=0,1,1,0,0,0,0,0"; do
    echo "== h0/c0 seed $seed  $e" >> $O
    if [ "$e" = default ]; then python3 $T 200 5 $seed fp32 2>/dev/null >> $O; else env $e python3 $T 200 5 $seed fp32 2>/dev/null >> $O; fi
  done
done
cat $O
