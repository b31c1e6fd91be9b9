set -x
mkdir -p gpurun_out/s38
export TMPDIR=/tmp
O=gpurun_out/s38
export TUTU_HIP_LIB=$PWD/tuturenderer_amd/libtutu_hip_census.so
for c in c3 c4; do for m in 0 1; do
TUTU_WIDE_COLLAPSE=$m timeout -k 10 300 python bench.py --config $c --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $O/bench_${c}_m$m.log 2> $O/bench_${c}_m$m.err
echo "== $c collapse $m"; grep census $O/bench_${c}_m$m.err | tail -2 | cut -c1-330
done; done
