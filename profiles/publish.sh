#!/bin/bash
# Copy the artefacts of `bash profiles/collect.sh <tag>` (merged back under gpurun_out/) into profiles/, replacing the
# ones of <old-tag>:   bash profiles/publish.sh <tag> [old-tag]
set -e
TAG=$1; OLD=$2
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/$TAG
[ -n "$OLD" ] && rm -f $R/profiles/${OLD}_*
cp $O/bench_line.json $R/profiles/${TAG}_bench_line.json
cp $O/bench_line_under_rocprof.json $R/profiles/${TAG}_bench_line_under_rocprof.json
cp $O/gpu_tests.log $R/profiles/${TAG}_gpu_tests.log
cp $O/scenes.jsonl $R/profiles/${TAG}_scenes.jsonl
cp $(find $O/rocprof -name "*kernel_stats.csv" | head -1) $R/profiles/${TAG}_bench_kernel_stats.csv
cp $(find $O/rocprof_exclusive -name "*kernel_stats.csv" | head -1) $R/profiles/${TAG}_bench_exclusive_kernel_stats.csv
python3 $R/profiles/pmc_summary.py $R/gpurun_out/pmc_$TAG > $R/profiles/${TAG}_pmc_summary.json
python3 $R/profiles/make_traffic.py $R/gpurun_out/pmc_$TAG > /dev/null
ls $R/profiles/${TAG}_*
