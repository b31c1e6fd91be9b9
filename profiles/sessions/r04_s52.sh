set -x
mkdir -p gpurun_out/s52
export TMPDIR=/tmp
O=gpurun_out/s52
run() { tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline --no-extras > $O/${c}_$tag.log 2>&1; python profiles/summarize_bench.py $O/${c}_$tag.log | sed "s#^$O/##" | cut -c1-50
python - <<PY
import json
d=json.loads([l for l in open('$O/${c}_$tag.log') if l.startswith('{')][-1]); print('   n_refs', d['config']['knobs'].get('n_refs'), 'wide_depth', d['config']['knobs'].get('wide_depth'))
PY
}
run base c4 1 TUTU_X=1
run m32 c4 1 TUTU_SPLIT_MAX=32
run m64 c4 1 TUTU_SPLIT_MAX=64
run m8 c4 1 TUTU_SPLIT_MAX=8
run g60 c4 1 TUTU_SPLIT_GAIN=0.6
run g80 c4 1 TUTU_SPLIT_GAIN=0.8
run m32g60 c4 1 TUTU_SPLIT_MAX=32 TUTU_SPLIT_GAIN=0.6
run base c5 3 TUTU_X=1
run m32 c5 3 TUTU_SPLIT_MAX=32
run g60 c5 3 TUTU_SPLIT_GAIN=0.6
run g80 c5 3 TUTU_SPLIT_GAIN=0.8
run m8 c5 3 TUTU_SPLIT_MAX=8
