set -x
mkdir -p gpurun_out/s53
export TMPDIR=/tmp
O=gpurun_out/s53
run() { tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline --no-extras > $O/${c}_$tag.log 2>&1; python profiles/summarize_bench.py $O/${c}_$tag.log | sed "s#^$O/##" | cut -c1-50
}
run base1 c4 1 TUTU_X=1
run early1 c4 1 TUTU_WIDE_EARLY=2
run base2 c4 1 TUTU_X=1
run early2 c4 1 TUTU_WIDE_EARLY=2
run noearly c5 3 TUTU_WIDE_EARLY=0
run base c5 3 TUTU_X=1
run noearly c3 3 TUTU_WIDE_EARLY=0
run base c3 3 TUTU_X=1
