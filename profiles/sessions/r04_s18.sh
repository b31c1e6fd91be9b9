set -x
mkdir -p gpurun_out/s18
export TMPDIR=/tmp
export TUTU_HIP_LIB=$PWD/tuturenderer_amd/libtutu_hip_census.so
for c in c5 c3 c4; do
timeout -k 10 300 python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/s18/bench_$c.log 2> gpurun_out/s18/bench_$c.err && grep census gpurun_out/s18/bench_$c.err | tail -4
done
