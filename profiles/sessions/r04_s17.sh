set -x
mkdir -p gpurun_out/s17
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_hip_frames.py tests/test_hip_wide.py -m gpu -q -s -k "native or flat or needle or c2" > gpurun_out/s17/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/s17/tests.log
grep -E "^\[|passed|failed|rc=|Error|error" gpurun_out/s17/tests.log | tail -40
for rep in 1 2; do
for sh in 0 1; do
TUTU_FLAT_SHARE=$sh timeout -k 10 300 python bench.py --config c2 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/s17/bench_c2_share${sh}_$rep.log 2>&1 && python profiles/summarize_bench.py gpurun_out/s17/bench_c2_share${sh}_$rep.log
done
done
for sh in 0 1; do
TUTU_FLAT_SHARE=$sh timeout -k 10 300 python bench.py --config c1 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/s17/bench_c1_share${sh}.log 2>&1 && python profiles/summarize_bench.py gpurun_out/s17/bench_c1_share${sh}.log
done
