set -x
mkdir -p gpurun_out/s33
export TMPDIR=/tmp
timeout -k 10 1000 python tests/tools/ray_parity.py 1048576 > gpurun_out/s33/ray_parity.log 2>&1; echo "rc=$?"
tail -14 gpurun_out/s33/ray_parity.log | cut -c1-220
