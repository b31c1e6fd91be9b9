# round 5, session 25: the whole GPU suite with exact_sum tests and atan2f; then ray parity tool
set -x
O=gpurun_out/s25; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -q -s -x > $O/gpu_tests.log 2>&1; echo "tests rc=$?"
grep "passed\|failed\|Error\|assert" $O/gpu_tests.log | cut -c1-330 | head -40
grep "exact_sum\]" $O/gpu_tests.log | cut -c1-200
