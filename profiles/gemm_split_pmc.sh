#!/bin/bash
# PMC passes over tools/gemm_bench.py on the step's two large main-stream shapes: the fp32-MFMA kernels (variant 8) against the split
# kernels (2: 128 x 128, 7: stream-K 256 x 256): how busy the matrix pipe is, what the split costs in VALU instructions, LDS activity
# and bank conflicts.  Separate passes per counter set (--kernel-trace --pmc only).
# usage (on the GPU box, through gpurun): profiles/gemm_split_pmc.sh <tag>
set -o pipefail
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
S="--shape 12800,7168,1792,0,1 --shape 12800,1792,7168,0,0"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  for v in 8 2 7 9; do
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/prof_${tag}_gsp_${v}_$i -- python3 $R/speech-separation_amd/tools/gemm_bench.py --variant $v $S > /dev/null 2> $O/prof_${tag}_gsp_${v}_$i.err || echo "pass $i variant $v failed"
  done
done
python3 - <<PY
import csv, glob, collections, re
print("# rocprofv3 PMC over tools/gemm_bench.py (12 launches per shape), per launch averages; shapes: NT 12800 x 7168 x 1792 (projection), NN 12800 x 1792 x 7168 (data gradient)")
print("# variant 8 = fp32-MFMA kernels (stream-K 256 x 256), 2 = split 128 x 128, 7 = split stream-K 256 x 256, 9 = split once while staging 256 x 128")
for v in (8, 2, 7, 9):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob("$O/prof_${tag}_gsp_%d_*/**/*counter_collection.csv" % v, recursive=True)):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "gemm_f32" not in k:
                continue
            m = re.search(r"gemm_f32_kernel\w*<[^>]*>", k)
            acc[m.group(0) if m else k[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name, d in sorted(acc.items()):
        print("variant %d  %s" % (v, name))
        for c, vals in sorted(d.items()):
            print("   %-30s %16.0f  (n=%d)" % (c, sum(vals) / len(vals), len(vals)))
        busy, mf = d.get("SQ_BUSY_CYCLES"), d.get("SQ_VALU_MFMA_BUSY_CYCLES")
        if busy and mf:
            print("   matrix pipe busy: SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES = %.3f (the guide's normalisation applies: see profiles/r04_lstm_pmc.txt)" % (sum(mf) / len(mf) / (sum(busy) / len(busy))))
PY
