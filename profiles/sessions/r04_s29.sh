set -x
mkdir -p gpurun_out/s29
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_wide.py tests/test_hip_frames.py -m gpu -x -q -k "flat or needle or c2" > gpurun_out/s29/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/s29/tests.log
tail -3 gpurun_out/s29/tests.log
O=gpurun_out/s29
for rep in 1 2 3; do for lib in prev head x; do
  if [ $lib = x ]; then L=$PWD/tuturenderer_amd/libtutu_hip.so; else L=$PWD/tuturenderer_amd/libtutu_hip_$lib.so; fi
  TUTU_HIP_LIB=$L timeout -k 10 300 python bench.py --config c2 --steps 4 --warmup 1 --no-cpu-baseline > $O/${lib}_c2_r$rep.log 2>&1
  python profiles/summarize_bench.py $O/${lib}_c2_r$rep.log | sed "s#^$O/##" | cut -c1-200
done; done
for lib in head x; do
  if [ $lib = x ]; then L=$PWD/tuturenderer_amd/libtutu_hip.so; else L=$PWD/tuturenderer_amd/libtutu_hip_$lib.so; fi
  TUTU_HIP_LIB=$L timeout -k 10 300 python bench.py --config c1 --steps 6 --warmup 2 --no-cpu-baseline > $O/${lib}_c1.log 2>&1
  python profiles/summarize_bench.py $O/${lib}_c1.log | sed "s#^$O/##" | cut -c1-200
done
