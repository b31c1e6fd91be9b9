#!/bin/bash
# One gpurun call that refreshes everything under profiles/ for a tag:  bash profiles/collect.sh r02_final
# (GPU tests log; bench lines of all five configs with the reference CPU baseline; rocprofv3 kernel stats of the default
#  bench command, overlapped and with one pass in flight; PMC passes for c2 at the benchmarked pass size)
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/gpu_tests.log
for c in c2 c1 c3 c5; do
  timeout -k 10 400 python bench.py --config $c --steps 3 --warmup 1 > $O/bench_$c.log 2>&1; echo "bench $c rc=$?"
done
timeout -k 10 400 python bench.py --config c4 --steps 1 --warmup 1 > $O/bench_c4.log 2>&1; echo "bench c4 rc=$?"
for c in c1 c2 c3 c4 c5; do grep "^{" $O/bench_$c.log > $O/bench_line_$c.json; done
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.log 2>&1); echo "rocprof rc=$?"
grep "^{" $O/bench_under_rocprof.log > $O/bench_line_under_rocprof.json
# the same command with ONE pass in flight (kernels run one at a time: exclusive per-launch durations, cf. the bench line's roofline)
(cd /tmp && export TUTU_SETS=1 && rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_exclusive -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_exclusive_under_rocprof.log 2>&1); echo "rocprof exclusive rc=$?"
bash profiles/run_pmc.sh $TAG c2 76 19 > $O/pmc.log 2>&1; echo "pmc rc=$?"
python profiles/summarize_bench.py $O/bench_c*.log
