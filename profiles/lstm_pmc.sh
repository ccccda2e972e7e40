#!/bin/bash
# PMC passes over tools/lstm_bench.py: the three forward recurrence kernels (plain fp32 MFMA + flags, the r03 tagged hand-off, the
# split-3 product) and the backward kernel, stand-alone, T=400 B=32 H=896.  Separate passes per counter set (--kernel-trace --pmc only).
# usage (on the GPU box, through gpurun): profiles/lstm_pmc.sh <tag>
set -o pipefail
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/prof_${tag}_lstm_$i -- python3 $R/speech-separation_amd/tools/lstm_bench.py --rounds 1 \
      --fwd "0,1,1,0,0,0,0,0;0,1,1,0,0,0,8,1;0,1,1,0,0,0,0,0,1" --bwd "0,1" > $O/prof_${tag}_lstm_$i.log 2> $O/prof_${tag}_lstm_$i.err || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
order = collections.defaultdict(int)
for f in sorted(glob.glob("$O/prof_${tag}_lstm_*/**/*counter_collection.csv", recursive=True)):
    seen = collections.defaultdict(int)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "lstm_fwd_kernel" in k:
            # launches alternate plain, tagged, split-3 (the tool's order); the split-3 instantiation has its own template flag
            name = "fwd split-3 (S3)" if "Lb0ELi8ELb1" in k.replace(" ", "") or ", true, false>" in k else None
            if name is None:
                key = (f, row["Counter_Name"])
                n = seen[key]; seen[key] += 1
                name = "fwd plain fp32 MFMA, flags" if n % 2 == 0 else "fwd plain fp32 MFMA, tagged hand-off"
        elif "lstm_bwd_kernel" in k:
            name = "bwd fp32"
        else:
            continue
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("# rocprofv3 PMC over tools/lstm_bench.py --rounds 1 (T=400, B=32, H=896: 224 workgroups of 512 threads; per launch, averages)")
for name, d in sorted(acc.items()):
    print(name)
    for c, v in sorted(d.items()):
        print("   %-28s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CYCLES" in d:
        m, b = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(d["SQ_VALU_MFMA_BUSY_CYCLES"]), sum(d["SQ_BUSY_CYCLES"]) / len(d["SQ_BUSY_CYCLES"])
        print("   MFMA busy / SQ busy cycles  %.3f" % (m / b))
    if "SQ_LDS_BANK_CONFLICT" in d and "SQ_LDS_IDX_ACTIVE" in d:
        print("   LDS bank conflict share      %.3f" % (sum(d["SQ_LDS_BANK_CONFLICT"]) / max(1.0, sum(d["SQ_LDS_IDX_ACTIVE"]))))
PY
