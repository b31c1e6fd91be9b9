set -x
mkdir -p gpurun_out/s4
timeout -k 10 200 python tests/tools/build_probe.py > gpurun_out/s4/build_probe.log 2>&1
cat gpurun_out/s4/build_probe.log
timeout -k 10 300 python -m pytest tests/test_hip_wide.py -m gpu -x -q -s -k "million or device_built" > gpurun_out/s4/tests.log 2>&1
tail -8 gpurun_out/s4/tests.log
for c in c2 c3 c5; do timeout -k 10 300 python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/s4/bench_$c.log 2>&1; python profiles/summarize_bench.py gpurun_out/s4/bench_$c.log; done
timeout -k 10 300 python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/s4/bench_c4.log 2>&1; python profiles/summarize_bench.py gpurun_out/s4/bench_c4.log
