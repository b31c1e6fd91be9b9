mkdir -p gpurun_out/s40
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -q -s -k "per_sample" > gpurun_out/s40/tests.log 2>&1; echo "tests rc=$?"
grep -E "worst|passed|failed|Error" gpurun_out/s40/tests.log | cut -c1-250
