set -x
mkdir -p gpurun_out/s37
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q -s > gpurun_out/s37/gpu_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/s37/gpu_tests.log
grep -E "passed|failed|rc=" gpurun_out/s37/gpu_tests.log | tail -3
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout -k 10 300 python bench.py > gpurun_out/s37/default_bench.json 2> gpurun_out/s37/default_bench.err; echo "bench rc=$?"; python profiles/summarize_bench.py gpurun_out/s37/default_bench.json | cut -c1-200
