set -x
mkdir -p gpurun_out/s26
export TMPDIR=/tmp
run() { # tag config steps env...
tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/s26/bench_${c}_$tag.log 2>gpurun_out/s26/bench_${c}_$tag.err && python profiles/summarize_bench.py gpurun_out/s26/bench_${c}_$tag.log | cut -c1-80
}
run base c2 8 TUTU_SHADE_BPC=0
run sb2 c2 8 TUTU_SHADE_BPC=2
run sb3 c2 8 TUTU_SHADE_BPC=3
run tb3 c2 8 TUTU_TRACE_BPC=3
run tb2 c2 8 TUTU_TRACE_BPC=2
run sb2_tb3 c2 8 TUTU_SHADE_BPC=2 TUTU_TRACE_BPC=3
run base2 c2 8 TUTU_SHADE_BPC=0
run sets3 c2 8 TUTU_SETS=3
run spp32 c2 8 TUTU_SHADE_BPC=0 
