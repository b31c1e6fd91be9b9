set -x
mkdir -p gpurun_out/s46
export TMPDIR=/tmp
O=gpurun_out/s46
for rep in 1 2; do for v in head x; do
  if [ $v = x ]; then L=$PWD/tuturenderer_amd/libtutu_hip.so; else L=$PWD/tuturenderer_amd/libtutu_hip_$v.so; fi
  echo "== $v cornell"; TUTU_HIP_LIB=$L timeout -k 10 200 python profiles/bench_integrators.py --steps 4 --no-cpu 2>/dev/null | grep "^{" | cut -c1-75
  echo "== $v veach"; TUTU_HIP_LIB=$L timeout -k 10 200 python profiles/bench_integrators.py --steps 4 --no-cpu --scene veach_room --width 800 --height 600 2>/dev/null | grep "^{" | cut -c1-75
done; done
timeout -k 10 600 python -m pytest tests/test_hip_integrators.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
