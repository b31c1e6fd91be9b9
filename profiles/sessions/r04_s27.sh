set -x
mkdir -p gpurun_out/s27
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_wide.py tests/test_hip_frames.py -m gpu -x -q -k "flat or needle or c2 or native" > gpurun_out/s27/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/s27/tests.log
tail -3 gpurun_out/s27/tests.log
bash profiles/ab.sh $PWD/tuturenderer_amd/libtutu_hip_prev.so $PWD/tuturenderer_amd/libtutu_hip.so 3 c2 c1
