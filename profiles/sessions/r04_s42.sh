set -x
mkdir -p gpurun_out/s42
export TMPDIR=/tmp
O=gpurun_out/s42
for rep in 1 2; do for v in head prio3; do
  L=$PWD/tuturenderer_amd/libtutu_hip_$v.so
  TUTU_HIP_LIB=$L timeout -k 10 300 python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/${v}_c3_r$rep.log 2>&1
  python profiles/summarize_bench.py $O/${v}_c3_r$rep.log | sed "s#^$O/##" | cut -c1-60
done; done
for v in head prio3 head prio3; do
  L=$PWD/tuturenderer_amd/libtutu_hip_$v.so
  TUTU_HIP_LIB=$L timeout -k 10 300 python bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/${v}_c4.log 2>&1
  python profiles/summarize_bench.py $O/${v}_c4.log | sed "s#^$O/##" | cut -c1-60
done
for rep in 1 2 3; do for v in head prio3; do
  L=$PWD/tuturenderer_amd/libtutu_hip_$v.so
  TUTU_HIP_LIB=$L timeout -k 10 300 python bench.py --config c2 --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $O/${v}_c2_r$rep.log 2>&1
  python profiles/summarize_bench.py $O/${v}_c2_r$rep.log | sed "s#^$O/##" | cut -c1-60
done; done
