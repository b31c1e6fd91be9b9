set -x
mkdir -p gpurun_out/s8
timeout -k 10 900 python -m pytest tests/test_hip_wide.py tests/test_hip_parity.py tests/test_hip_frames.py tests/test_host_frontend.py tests/test_boundary_reference.py -m gpu -q -s -k "not c3_ and not c4_ and not c5_ and not needles and not million" > gpurun_out/s8/tests.log 2>&1
tail -12 gpurun_out/s8/tests.log
grep -E "^\[flat|^\[tree" gpurun_out/s8/tests.log
for f in 1 0 1; do TUTU_FLAT=$f timeout -k 10 300 python bench.py --config c2 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/s8/bench_c2_flat$f.log 2>&1; echo flat=$f; python profiles/summarize_bench.py gpurun_out/s8/bench_c2_flat$f.log; done
timeout -k 10 300 python bench.py --config c1 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/s8/bench_c1.log 2>&1; python profiles/summarize_bench.py gpurun_out/s8/bench_c1.log
