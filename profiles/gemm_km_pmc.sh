#!/bin/bash
# PMC passes over tools/gemm_bench.py --km: K-major (transposed LDS reads) vs NT form of the bf16 GEMM, one square shape.
# usage (on the GPU box, through gpurun): profiles/gemm_km_pmc.sh <tag>
set -o pipefail
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/prof_${tag}_km_$i -- python3 $R/speech-separation_amd/tools/gemm_bench.py --km --only 7 > $O/prof_${tag}_km_$i.log 2> $O/prof_${tag}_km_$i.err || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/prof_${tag}_km_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "gemm_bf16_nt_kernel" not in k:
            continue
        name = "K-major" if "Lb1E" in k else "NT"
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, d in acc.items():
    print(name, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())})
PY
