set -x
mkdir -p gpurun_out/s21
export TMPDIR=/tmp
run() { # tag config steps env...
tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline > gpurun_out/s21/bench_${c}_$tag.log 2>gpurun_out/s21/bench_${c}_$tag.err && python profiles/summarize_bench.py gpurun_out/s21/bench_${c}_$tag.log
}
for c in c5 c3; do
run off $c 4 TUTU_WIDE_STAGE=0
run on8 $c 4 TUTU_WIDE_STAGE=1
run on4 $c 4 TUTU_STAGE_MIN=4
run on8_e0 $c 4 TUTU_WIDE_EARLY=0
done
run off c4 1 TUTU_WIDE_STAGE=0
run on8 c4 1 TUTU_WIDE_STAGE=1
run on4 c4 1 TUTU_STAGE_MIN=4
run census c5 1 TUTU_HIP_LIB=$PWD/tuturenderer_amd/libtutu_hip_census.so
grep census gpurun_out/s21/bench_c5_census.err | head -2
run census c4 1 TUTU_HIP_LIB=$PWD/tuturenderer_amd/libtutu_hip_census.so
grep census gpurun_out/s21/bench_c4_census.err | head -2
