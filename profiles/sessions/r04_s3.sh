set -x
mkdir -p gpurun_out/s3
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q -s > gpurun_out/s3/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/s3/tests.log
grep -E "^\[|passed|failed|rc=|Error|error" gpurun_out/s3/tests.log | tail -60
