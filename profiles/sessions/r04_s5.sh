set -x
mkdir -p gpurun_out/s5
timeout -k 10 600 python -m pytest tests/test_hip_wide.py tests/test_hip_parity.py tests/test_hip_frames.py -m gpu -x -q -s -k "flat_scan or closest_and_any or per_sample or c2_ or headline or cold_start or rccl or two_ranks" > gpurun_out/s5/tests.log 2>&1
tail -15 gpurun_out/s5/tests.log
for f in 1 0 1 0; do TUTU_FLAT=$f timeout -k 10 300 python bench.py --config c2 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/s5/bench_c2_flat$f.log 2>&1; echo flat=$f; python profiles/summarize_bench.py gpurun_out/s5/bench_c2_flat$f.log; done
TUTU_FLAT=1 timeout -k 10 300 python bench.py --config c1 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/s5/bench_c1.log 2>&1; python profiles/summarize_bench.py gpurun_out/s5/bench_c1.log
