set -x
mkdir -p gpurun_out/s35
export TMPDIR=/tmp
O=gpurun_out/s35
echo skip tests
run() { # tag config steps env...
tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline > $O/bench_${c}_$tag.log 2>$O/bench_${c}_$tag.err && python profiles/summarize_bench.py $O/bench_${c}_$tag.log | cut -c1-230
}
for rep in 1 2; do
for c in c5 c3; do
run pairs_$rep $c 4 TUTU_WIDE_COLLAPSE=0
run fill_$rep $c 4 TUTU_WIDE_COLLAPSE=2
run greedy_$rep $c 4 TUTU_WIDE_COLLAPSE=1
done
run pairs_$rep c4 1 TUTU_WIDE_COLLAPSE=0
run fill_$rep c4 1 TUTU_WIDE_COLLAPSE=2
run greedy_$rep c4 1 TUTU_WIDE_COLLAPSE=1
done
