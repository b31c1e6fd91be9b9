#!/bin/bash
# Copy the artefacts of `bash profiles/collect.sh <tag>` (merged back under gpurun_out/) into profiles/:  bash profiles/publish.sh <tag>
set -e
TAG=$1
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/$TAG
for c in c1 c2 c3 c4 c5; do cp $O/bench_line_$c.json $R/profiles/${TAG}_bench_line_$c.json; done
cp $O/bench_line_under_rocprof.json $R/profiles/${TAG}_bench_line_under_rocprof.json
cp $O/gpu_tests.log $R/profiles/${TAG}_gpu_tests.log
cp $(find $O/rocprof -name "*kernel_stats.csv" | head -1) $R/profiles/${TAG}_bench_kernel_stats.csv
cp $(find $O/rocprof_exclusive -name "*kernel_stats.csv" | head -1) $R/profiles/${TAG}_bench_exclusive_kernel_stats.csv
python3 $R/profiles/pmc_summary.py $R/gpurun_out/pmc_${TAG}_c2 > $R/profiles/${TAG}_pmc_summary_c2.json
python3 $R/profiles/make_traffic.py $R/gpurun_out/pmc_${TAG}_c2 $TAG c2 > /dev/null
ls $R/profiles/${TAG}_*
