#!/bin/bash
# One gpurun call that refreshes everything under profiles/ for a tag:  bash profiles/collect.sh r01_final
# (GPU tests log, bench line with reference CPU baseline, rocprofv3 kernel stats of the same bench command, PMC passes)
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/gpu_tests.log
timeout -k 10 600 python bench.py --steps 3 --warmup 1 > $O/bench.log 2>&1; echo "bench rc=$?"
grep "^{" $O/bench.log > $O/bench_line.json
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1); echo "rocprof rc=$?"
grep "^{" $O/bench_under_rocprof.log > $O/bench_line_under_rocprof.json
# the same command with ONE pass in flight (kernels run one at a time: exclusive per-launch durations, cf. roofline.exclusive)
(cd /tmp && export TUTU_SETS=1 && rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_exclusive -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_exclusive_under_rocprof.log 2>&1); echo "rocprof exclusive rc=$?"
mkdir -p $R/gpurun_out/pmc_$TAG
bash profiles/run_pmc.sh $TAG 64 > $O/pmc.log 2>&1; echo "pmc rc=$?"
timeout -k 10 600 python tests/tools/bench_scenes.py cornell veach veach_slight bunny broom cornell_textured cornell_spheres > $O/scenes.jsonl 2>&1; echo "scenes rc=$?"
tail -1 $O/gpu_tests.log; cut -c1-300 $O/bench_line.json
