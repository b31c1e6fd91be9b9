set -x
mkdir -p gpurun_out/s51
export TMPDIR=/tmp
O=gpurun_out/s51
timeout -k 10 900 python -m pytest tests/test_hip_wide.py tests/test_hip_frames.py tests/test_hip_parity.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
run() { tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline > $O/${c}_$tag.log 2>&1; python profiles/summarize_bench.py $O/${c}_$tag.log | sed "s#^$O/##" | cut -c1-200
python - <<PY
import json
d=json.loads([l for l in open('$O/${c}_$tag.log') if l.startswith('{')][-1]); print('   wide_greedy', d['config']['knobs'].get('wide_greedy'), 'crc', d['frame']['crc32'])
PY
}
for rep in 1 2; do
run pairs_$rep c5 3 TUTU_WIDE_COLLAPSE=0
run auto_$rep c5 3 TUTU_X=1
run pairs_$rep c3 3 TUTU_WIDE_COLLAPSE=0
run auto_$rep c3 3 TUTU_X=1
done
run pairs c4 1 TUTU_WIDE_COLLAPSE=0
run auto c4 1 TUTU_X=1
