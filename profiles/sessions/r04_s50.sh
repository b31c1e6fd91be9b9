set -x
mkdir -p gpurun_out/s50
export TMPDIR=/tmp
O=gpurun_out/s50
run() { tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline --no-extras > $O/${c}_$tag.log 2>&1; python profiles/summarize_bench.py $O/${c}_$tag.log | sed "s#^$O/##" | cut -c1-50
}
for rep in 1 2; do
for rf in 24 20 16 12; do
run rf${rf}_$rep c5 3 TUTU_REFILL_MIN=$rf
run rf${rf}_$rep c3 3 TUTU_REFILL_MIN=$rf
done; done
for rf in 24 16 12 24 16; do
run rf${rf}_x$RANDOM c4 1 TUTU_REFILL_MIN=$rf
done
