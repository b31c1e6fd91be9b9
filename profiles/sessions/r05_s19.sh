# round 5, session 19: the driver's own sequence on the final tree: GPU suite, smoke(), default bench line
set -x
O=gpurun_out/s19; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -q -s > $O/gpu_tests.log 2>&1; echo "tests rc=$?" | tee -a $O/gpu_tests.log; tail -3 $O/gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 400 python bench.py > $O/bench_default.log 2>&1; python profiles/summarize_bench.py $O/bench_default.log | cut -c1-300
