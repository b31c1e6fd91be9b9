# round 5, session 14: 21 M path-like rays over the eight golden scenes against the CPU restatement on the final tree (the memory-resident
# scenes now walk the eight-wide tree), the same with the eight-wide walk forced everywhere it exists, and the new far-origin test
set -x
O=gpurun_out/s14; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python tests/tools/ray_parity.py > $O/ray_parity.log 2>&1; echo "rc=$?"; cat $O/ray_parity.log | cut -c1-200
timeout -k 10 600 python -m pytest tests/test_hip_wide.py -m gpu -x -q -k "far_origin or needles" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
