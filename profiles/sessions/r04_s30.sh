set -x
mkdir -p gpurun_out/s30
export TMPDIR=/tmp
O=gpurun_out/s30
for rep in 1 2 3; do for lib in head x; do
  if [ $lib = x ]; then L=$PWD/tuturenderer_amd/libtutu_hip.so; else L=$PWD/tuturenderer_amd/libtutu_hip_$lib.so; fi
  TUTU_HIP_LIB=$L timeout -k 10 300 python bench.py --config c2 --steps 4 --warmup 1 --no-cpu-baseline > $O/${lib}_c2_r$rep.log 2>&1
  python profiles/summarize_bench.py $O/${lib}_c2_r$rep.log | sed "s#^$O/##" | cut -c1-200
done; done
