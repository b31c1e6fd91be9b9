set -x
mkdir -p gpurun_out/s48
export TMPDIR=/tmp
O=gpurun_out/s48
for rep in 1 2; do for v in head x; do
  if [ $v = x ]; then L=$PWD/tuturenderer_amd/libtutu_hip.so; else L=$PWD/tuturenderer_amd/libtutu_hip_$v.so; fi
  for c in c5 c3; do
  TUTU_HIP_LIB=$L timeout -k 10 300 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > $O/${v}_${c}_r$rep.log 2>&1
  python profiles/summarize_bench.py $O/${v}_${c}_r$rep.log | sed "s#^$O/##" | cut -c1-200
  done
done; done
for v in head x; do
  if [ $v = x ]; then L=$PWD/tuturenderer_amd/libtutu_hip.so; else L=$PWD/tuturenderer_amd/libtutu_hip_$v.so; fi
  TUTU_HIP_LIB=$L timeout -k 10 300 python bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/${v}_c4.log 2>&1
  python profiles/summarize_bench.py $O/${v}_c4.log | sed "s#^$O/##" | cut -c1-60
done
