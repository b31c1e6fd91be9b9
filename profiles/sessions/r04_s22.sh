set -x
mkdir -p gpurun_out/s22
export TMPDIR=/tmp
run() { # tag config steps env...
tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline > gpurun_out/s22/bench_${c}_$tag.log 2>gpurun_out/s22/bench_${c}_$tag.err && python profiles/summarize_bench.py gpurun_out/s22/bench_${c}_$tag.log
}
for c in c5 c3; do
run base $c 4 TUTU_LEAF_STEPS=2
run ls3 $c 4 TUTU_LEAF_STEPS=3
run ls4 $c 4 TUTU_LEAF_STEPS=4
run ls4_la16 $c 4 TUTU_LEAF_STEPS=4 TUTU_LEAF_AGAIN=16
run ls4_la32 $c 4 TUTU_LEAF_STEPS=4 TUTU_LEAF_AGAIN=32
run st16 $c 4 TUTU_STUCK_MAX=16
run st24 $c 4 TUTU_STUCK_MAX=24
run st24_ls4 $c 4 TUTU_STUCK_MAX=24 TUTU_LEAF_STEPS=4
run st24_is6 $c 4 TUTU_STUCK_MAX=24 TUTU_WIDE_INNER_STEPS=6 TUTU_WIDE_INNER_STEPS_ANY=6
done
run base c4 1 TUTU_LEAF_STEPS=2
run ls4 c4 1 TUTU_LEAF_STEPS=4
run st24 c4 1 TUTU_STUCK_MAX=24
run st24_is6 c4 1 TUTU_STUCK_MAX=24 TUTU_WIDE_INNER_STEPS=6 TUTU_WIDE_INNER_STEPS_ANY=6
