# round 5, session 41: what the driver runs at round end, on the final tree: the GPU suite, smoke(), the default bench line
O=gpurun_out/s41; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -1 $O/gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout -k 10 600 python bench.py > $O/bench_default.log 2>&1; echo "bench rc=$?"; grep "^{" $O/bench_default.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'scaling', 'vs_baseline', 'dtype', 'data')}, d['roofline']['kernel'], round(d['roofline']['frac'], 3), d['cpu_baseline'])"
