set -x
mkdir -p gpurun_out/s36
export TMPDIR=/tmp
O=gpurun_out/s36
run() { # tag config steps env...
tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline --no-extras > $O/bench_${c}_$tag.log 2>$O/bench_${c}_$tag.err && python profiles/summarize_bench.py $O/bench_${c}_$tag.log | cut -c1-60
grep "wide collapse" $O/bench_${c}_$tag.err | head -1
}
run auto c5 3 TUTU_BUILD_TIMING=1
run auto c3 3 TUTU_BUILD_TIMING=1
run auto c4 1 TUTU_BUILD_TIMING=1
timeout -k 10 600 python -m pytest tests/test_hip_wide.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
