#!/bin/bash
# Install what profiles/collect.sh <tag> left in gpurun_out/ as the tracked files of that tag (run here, after the gpurun call):
#   profiles/<tag>_{bench,bench_bf16_3spk}.json, <tag>_{,bf16_}kernel_stats.csv, <tag>_{,bf16_}summary.txt, pmc_traffic.json
# usage: profiles/install.sh <tag>
set -e
T=$1; O=gpurun_out
cd "$(dirname "$0")/.."
cp $O/bench_$T.json profiles/${T}_bench.json
cp $O/bench_${T}_bf16.json profiles/${T}_bench_bf16_3spk.json
for v in ragged ragged_padded_rows plain_fp32_mfma rsh_4spk bf16_ragged; do [ -f $O/bench_${T}_$v.json ] && cp $O/bench_${T}_$v.json profiles/${T}_bench_$v.json; done
cp $(ls $O/prof_${T}_trace/runc/*kernel_stats.csv | tail -1) profiles/${T}_kernel_stats.csv
cp $(ls $O/prof_${T}_bf16_trace/runc/*kernel_stats.csv | tail -1) profiles/${T}_bf16_kernel_stats.csv
python profiles/summarize.py $T $O/prof_${T}_trace $O/prof_${T}_f32_FETCH_SIZE $O/prof_${T}_f32_WRITE_SIZE > profiles/${T}_summary.txt
python profiles/summarize.py ${T}_bf16 $O/prof_${T}_bf16_trace $O/prof_${T}_bf16_FETCH_SIZE $O/prof_${T}_bf16_WRITE_SIZE > profiles/${T}_bf16_summary.txt
cp $O/pmc_traffic_$T.json profiles/pmc_traffic.json   # made on the box from the same sources (kernel_source_id), before the bench lines
tail -3 $O/pmc_traffic_$T.txt
