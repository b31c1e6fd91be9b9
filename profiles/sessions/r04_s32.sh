set -x
mkdir -p gpurun_out/s32
export TMPDIR=/tmp
O=gpurun_out/s32
timeout -k 10 300 env TUTU_FLAT_SHARE_ANY=1 python -m pytest tests/test_hip_wide.py -m gpu -x -q -k "flat" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
for rep in 1 2 3; do for v in head x0 x1; do
  case $v in
    head) L=$PWD/tuturenderer_amd/libtutu_hip_head.so; E=0;;
    x0) L=$PWD/tuturenderer_amd/libtutu_hip.so; E=0;;
    x1) L=$PWD/tuturenderer_amd/libtutu_hip.so; E=1;;
  esac
  TUTU_FLAT_SHARE_ANY=$E TUTU_HIP_LIB=$L timeout -k 10 300 python bench.py --config c2 --steps 4 --warmup 1 --no-cpu-baseline > $O/${v}_c2_r$rep.log 2>&1
  python profiles/summarize_bench.py $O/${v}_c2_r$rep.log | sed "s#^$O/##" | cut -c1-200
done; done
