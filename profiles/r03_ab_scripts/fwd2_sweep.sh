set -x
cd "$(dirname "$0")/.."
timeout -k 10 200 python -m pytest ../tests/test_gpu_kernels.py -x -q -m gpu -k "two_stream" 2>&1 | tail -3
V="0,1,1;0,1,1,0,0,0,8,1;0,1,1,0,0,0,12,1;0,1,1,0,0,0,16,1;0,1,1,0,0,0,20,1"
for lib in "" strict noprio ps4; do
  echo "== lib ${lib:-default}"
  if [ -n "$lib" ]; then export SEPKERN_LIB=$PWD/sepkern/libsepkern_$lib.so; else unset SEPKERN_LIB; fi
  timeout -k 10 200 python tools/lstm_bench.py --rounds 4 --fwd "$V" --bwd "0,1,0,0,0,0,31" 2>&1 | grep -v amdgpu.ids
done
export SEPKERN_LIB=$PWD/sepkern/libsepkern_stamps.so
timeout -k 10 200 python tools/lstm_stamps.py --dual 12,16 2>&1 | grep -v amdgpu.ids
