# round 5, session 2: first run of the eight-wide walk -- the forms test, then same-box A/B of TUTU_WIDE8=0 / 1 on c3, c5, c4
set -x
O=gpurun_out/s2; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_hip_wide.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -15 $O/tests.log
[ $rc -ne 0 ] && exit 1
run() { tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline > $O/${c}_$tag.log 2>&1 || { echo "bench failed"; tail -5 $O/${c}_$tag.log; exit 1; }
python profiles/summarize_bench.py $O/${c}_$tag.log | sed "s#^$O/##" | cut -c1-230
python - <<PY
import json
d=json.loads([l for l in open('$O/${c}_$tag.log') if l.startswith('{')][-1]); k=d['config']['knobs']; print('   wide8_tree', k.get('wide8_tree'), 'depth', k.get('wide8_depth'), 'nodes', k.get('wide8_nodes'), 'entries', k.get('wide8_entries'), 'bpc', k.get('trace_blocks_per_cu'), 'crc', d['frame']['crc32'])
PY
}
for rep in 1 2; do
run w4_$rep c3 3 TUTU_WIDE8=0
run w8_$rep c3 3 TUTU_WIDE8=1
run w4_$rep c5 3 TUTU_WIDE8=0
run w8_$rep c5 3 TUTU_WIDE8=1
done
run w4 c4 1 TUTU_WIDE8=0
run w8 c4 1 TUTU_WIDE8=1
