# round 5, session 42: exact_sum on the stand-in meshes' golden samples and on 2000 random samples of every BASELINE frame; ray parity on the final tree
O=gpurun_out/s42; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "stand_in or headline or config" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
timeout -k 10 600 python tests/tools/ray_parity.py > $O/ray_parity.log 2>&1; echo "ray parity rc=$?"; tail -4 $O/ray_parity.log
