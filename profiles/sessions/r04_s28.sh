set -x
mkdir -p gpurun_out/s28
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_wide.py -m gpu -x -q > gpurun_out/s28/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/s28/tests.log
tail -3 gpurun_out/s28/tests.log
P=$PWD/tuturenderer_amd/libtutu_hip_prev.so
run() { # tag config steps env...
tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline > gpurun_out/s28/bench_${c}_$tag.log 2>gpurun_out/s28/bench_${c}_$tag.err && python profiles/summarize_bench.py gpurun_out/s28/bench_${c}_$tag.log | cut -c1-200
}
for c in c5 c3; do
run prev $c 4 TUTU_HIP_LIB=$P
run s19 $c 4 TUTU_WIDE_LDS_STACK=19
run s14 $c 4 TUTU_WIDE_LDS_STACK=14
run s10 $c 4 TUTU_WIDE_LDS_STACK=10
run s7 $c 4 TUTU_WIDE_LDS_STACK=7
done
run prev c4 1 TUTU_HIP_LIB=$P
run s19 c4 1 TUTU_WIDE_LDS_STACK=19
run s14 c4 1 TUTU_WIDE_LDS_STACK=14
run s10 c4 1 TUTU_WIDE_LDS_STACK=10
run prev c2 4 TUTU_HIP_LIB=$P
run new c2 4 TUTU_WIDE_LDS_STACK=19
run prev2 c2 4 TUTU_HIP_LIB=$P
run new2 c2 4 TUTU_WIDE_LDS_STACK=19
