set -x
mkdir -p gpurun_out/s20
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_wide.py tests/test_hip_parity.py tests/test_hip_frames.py -m gpu -x -q > gpurun_out/s20/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/s20/tests.log
tail -5 gpurun_out/s20/tests.log
run() { # tag config steps env...
tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline > gpurun_out/s20/bench_${c}_$tag.log 2>&1 && python profiles/summarize_bench.py gpurun_out/s20/bench_${c}_$tag.log
}
for c in c5 c3; do
run off $c 4 TUTU_WIDE_STAGE=0
run on8 $c 4 TUTU_WIDE_STAGE=1
run on4 $c 4 TUTU_STAGE_MIN=4
run on16 $c 4 TUTU_STAGE_MIN=16
run on8_s15 $c 4 TUTU_WIDE_LDS_STACK=15
run on4_s15 $c 4 TUTU_WIDE_LDS_STACK=15 TUTU_STAGE_MIN=4
done
run off c4 1 TUTU_WIDE_STAGE=0
run on8 c4 1 TUTU_WIDE_STAGE=1
run on4_s15 c4 1 TUTU_WIDE_LDS_STACK=15 TUTU_STAGE_MIN=4
run on8_s15 c4 1 TUTU_WIDE_LDS_STACK=15
