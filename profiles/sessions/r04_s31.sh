set -x
mkdir -p gpurun_out/s31
export TMPDIR=/tmp
run() { # tag config steps env...
tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/s31/bench_${c}_$tag.log 2>gpurun_out/s31/bench_${c}_$tag.err && python profiles/summarize_bench.py gpurun_out/s31/bench_${c}_$tag.log | cut -c1-80
}
for rep in 1 2; do
run base_$rep c2 6 TUTU_SHADE_BPC=0
run sb2_$rep c2 6 TUTU_SHADE_BPC=2
run sb3_$rep c2 6 TUTU_SHADE_BPC=3
run tb6_$rep c2 6 TUTU_TRACE_BPC=6
run tb5_$rep c2 6 TUTU_TRACE_BPC=5
done
