#!/bin/bash
# HBM-side traffic of the fp32 GEMM kernels on the step's two large main-stream shapes: chosen-before (variant 4: 256 x 128 tiles),
# plain 256 x 256 (5), stream-K (6).  Separate --pmc passes; per-launch FETCH_SIZE / WRITE_SIZE printed by the python at the end.
# usage (GPU box): profiles/gemm_sk_pmc.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
S="--shape 12800,7168,1792,0,1 --shape 12800,1792,7168,0,0"
for v in 4 5 6; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/prof_gsk_${v}_$c -- python3 $R/speech-separation_amd/tools/gemm_bench.py --variant $v $S > /dev/null 2> $O/prof_gsk_${v}_$c.err || exit 1
  done
done
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out"
print("# per-launch HBM-side bytes (FETCH_SIZE x2 per the guide's gfx950 correction, WRITE_SIZE as read); algorithmic: projection 0.51 GB, data gradient 0.51 GB")
for v in (4, 5, 6):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob("%s/prof_gsk_%d_%s/**/*counter_collection.csv" % (O, v, c), recursive=True)[0]
        for r in csv.DictReader(open(f)):
            if "gemm_f32" in r["Kernel_Name"]:
                acc[__import__("re").search(r"gemm_f32_kernel\w*<[^>]*>", r["Kernel_Name"]).group(0)][c].append(float(r["Counter_Value"]))
    for k, e in acc.items():
        fe = sum(e["FETCH_SIZE"]) / len(e["FETCH_SIZE"]) * 1024 * 2 / 1e9
        wr = sum(e["WRITE_SIZE"]) / len(e["WRITE_SIZE"]) * 1024 / 1e9
        print("variant %d  %-42s launches %3d  fetch %.3f GB  write %.3f GB" % (v, k, len(e["FETCH_SIZE"]), fe, wr))
PY
