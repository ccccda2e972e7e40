#!/bin/bash
# r06, first GPU call: (1) what the hwmon sensors offer; (2) signed error of the split arithmetic, product build vs a build without
# the sign phases; (3) timing-only bound for a split-product backward recurrence (3 of 8 MFMAs: top layer only / all layers);
# (4) regression of the kernel / model tests on the new build; (5) the default bench line without the CPU baseline.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
S=$R/speech-separation_amd/sepkern
cd $R
{ for h in /sys/class/drm/card*/device/hwmon/hwmon*; do echo "== $h"; ls $h | tr '\n' ' '; echo; for f in power1_average power1_input power1_cap freq1_input; do [ -r $h/$f ] && echo "$f $(cat $h/$f)"; done; done; } > $O/r06_hwmon.txt 2>&1
mkdir -p /tmp/se_cache
timeout -k 10 400 python speech-separation_amd/tools/signed_error.py --cache /tmp/se_cache > $O/r06_signed_error.txt 2>&1 || { echo "signed_error failed"; tail -5 $O/r06_signed_error.txt; exit 1; }
SEPKERN_ALLOW_DIAGNOSTIC_LIB=1 SEPKERN_LIB=$S/libsepkern_noflip.so timeout -k 10 400 python speech-separation_amd/tools/signed_error.py --cache /tmp/se_cache > $O/r06_signed_error_noflip.txt 2>&1 || { echo "signed_error noflip failed"; tail -5 $O/r06_signed_error_noflip.txt; exit 1; }
OUT=$O/r06_bwd_split_top_layer.txt
: > $OUT
line() { python -c "
import json,sys
l=[x for x in sys.stdin.read().strip().splitlines() if x.startswith('{')]
if not l: print('$1: no line'); sys.exit(0)
d=json.loads(l[-1]); k=d['kernels']
print('%-10s %.3f ms/step  ' % ('$1', d['ms_per_step']) + '  '.join('%s %.2f' % (n, v['ms_per_step']) for n, v in k.items()))
"; }
for i in 1 2; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-aux --no-power-probe --steps 20 --warmup 3 2>/dev/null | line default | tee -a $OUT
  SEPKERN_ALLOW_DIAGNOSTIC_LIB=1 SEPKERN_LIB=$S/libsepkern_bwd38.so SEPKERN_BWD_DIAG=1 timeout -k 10 200 python bench.py --diagnostic --no-cpu-baseline --no-secondary --no-aux --no-power-probe --steps 20 --warmup 3 2>/dev/null | line top38 | tee -a $OUT
  SEPKERN_ALLOW_DIAGNOSTIC_LIB=1 SEPKERN_LIB=$S/libsepkern_bwd38.so SEPKERN_BWD_DIAG=2 timeout -k 10 200 python bench.py --diagnostic --no-cpu-baseline --no-secondary --no-aux --no-power-probe --steps 20 --warmup 3 2>/dev/null | line all38 | tee -a $OUT
done
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q > $O/r06a_tests.log 2>&1; echo "pytest rc $?" | tee -a $O/r06a_tests.log; tail -3 $O/r06a_tests.log
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/r06a_bench.json 2> $O/r06a_bench.err; echo "bench rc $?"; tail -c 600 $O/r06a_bench.json
