#!/bin/bash
# A/B of the bf16 configuration (BASELINE configs[3]) in ONE call on one device: in-kernel vs separate split-K reduction.
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  for ik in 1 0; do
    echo "== SEPKERN_BF16_SPLITK_INKERNEL=$ik (rep $rep)"
    SEPKERN_BF16_SPLITK_INKERNEL=$ik timeout -k 10 200 python bench.py --dtype bf16 --num-spk 3 --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k:v['ms_per_step'] for k,v in d['kernels'].items()})"
  done
done
echo "== gemm_bench --nt (in-kernel)"
cd speech-separation_amd
timeout -k 10 200 python tools/gemm_bench.py --nt 2>&1 | grep -v amdgpu.ids
echo "== gemm_bench --nt (separate reduce)"
SEPKERN_BF16_SPLITK_INKERNEL=0 timeout -k 10 200 python tools/gemm_bench.py --nt 2>&1 | grep -v amdgpu.ids
