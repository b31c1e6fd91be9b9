#!/bin/bash
# Copy the artefacts of `bash profiles/collect.sh <tag> <stage>` (merged back under gpurun_out/) into profiles/:
#   bash profiles/publish.sh <tag> pmc | lines
set -e
TAG=$1
STAGE=${2:-pmc}
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/$TAG
if [ "$STAGE" = "pmc" ]; then
  cp $O/gpu_tests.log $R/profiles/${TAG}_gpu_tests.log
  for c in c2 c3 c4 c5; do
    python3 $R/profiles/pmc_summary.py $R/gpurun_out/pmc_${TAG}_$c > $R/profiles/${TAG}_pmc_summary_$c.json
    python3 $R/profiles/make_traffic.py $R/gpurun_out/pmc_${TAG}_$c $TAG $c > /dev/null
  done
else
  cp $O/bench_line_under_rocprof.json $R/profiles/${TAG}_bench_line_under_rocprof.json
  cp $(find $O/rocprof -name "*kernel_stats.csv" | head -1) $R/profiles/${TAG}_bench_kernel_stats.csv
  cp $(find $O/rocprof_exclusive -name "*kernel_stats.csv" | head -1) $R/profiles/${TAG}_bench_exclusive_kernel_stats.csv
  for c in c1 c2 c3 c4 c5; do cp $O/bench_line_$c.json $R/profiles/${TAG}_bench_line_$c.json; done
  cp $O/shard_balance_c2.json $R/profiles/${TAG}_shard_balance_c2.json
  cp $O/shard_balance_c4.json $R/profiles/${TAG}_shard_balance_c4.json
  cp $O/integrators_bench.jsonl $R/profiles/${TAG}_integrators_bench.jsonl
fi
ls $R/profiles/${TAG}_*
