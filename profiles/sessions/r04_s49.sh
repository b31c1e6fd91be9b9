set -x
mkdir -p gpurun_out/s49
export TMPDIR=/tmp
O=gpurun_out/s49
run() { tag=$1; c=$2; shift 2
env "$@" timeout -k 10 300 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/${c}_$tag.log 2>&1; python profiles/summarize_bench.py $O/${c}_$tag.log | sed "s#^$O/##" | cut -c1-50
}
for c in c5 c3; do
run base $c TUTU_REFILL_MIN=24
run rf16 $c TUTU_REFILL_MIN=16
run rf32 $c TUTU_REFILL_MIN=32
run is3 $c TUTU_WIDE_INNER_STEPS=3
run is5 $c TUTU_WIDE_INNER_STEPS=5
run isa3 $c TUTU_WIDE_INNER_STEPS_ANY=3
run isa5 $c TUTU_WIDE_INNER_STEPS_ANY=5
run la16 $c TUTU_LEAF_AGAIN=16
run la32 $c TUTU_LEAF_AGAIN=32
run base2 $c TUTU_REFILL_MIN=24
done
