#!/bin/bash
# fp32 GEMM by the three-way bf16 split with SIX piece products (variant 2) against the shipped fp32-MFMA kernels (variant 0: LDS-DMA /
# stream-K) and the nine-product form (libsepkern_nine.so, -DSK_SPLIT_NINE), on the training step's shapes; then sustained with power.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05_gemm_split6.txt
T=$R/speech-separation_amd/tools
: > $O
for rep in 1 2; do
  echo "== variant 0 (fp32 MFMA kernels as shipped)" >> $O; python3 $T/gemm_bench.py --variant 0 >> $O 2>&1 || exit 1
  echo "== variant 2 six products" >> $O; python3 $T/gemm_bench.py --variant 2 >> $O 2>&1 || exit 1
  echo "== variant 2 nine products (diagnostic build)" >> $O; SEPKERN_LIB=$R/speech-separation_amd/sepkern/libsepkern_nine.so python3 $T/gemm_bench.py --variant 2 >> $O 2>&1 || exit 1
done
echo "== sustained, with rocm-smi (six products in the default library)" >> $O; python3 $T/gemm_power.py 5 >> $O 2>&1
cat $O
