# round 5, session 23: the whole GPU suite on the tree with the C library's functions on the device and the tightened bars
set -x
O=gpurun_out/s23; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -q -s -x > $O/gpu_tests.log 2>&1; echo "tests rc=$?"
grep "passed\|failed\|Error\|assert" $O/gpu_tests.log | cut -c1-330 | head -40
grep "differing last bit\|diverged" $O/gpu_tests.log | cut -c1-260
