# round 5, session 15: bigger passes -- 336 Mi path slots in flight (134 GB) instead of 168 Mi: fewer, larger launches
O=gpurun_out/s15; mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; c=$2; st=$3; shift 3
timeout -k 10 300 python bench.py --config $c --steps $st --warmup 2 --no-cpu-baseline --no-extras "$@" > $O/${c}_$tag.log 2>&1 || { echo "bench failed $c $tag"; tail -3 $O/${c}_$tag.log; return; }
python - <<PY
import json
d=json.loads([l for l in open('$O/${c}_$tag.log') if l.startswith('{')][-1]); print(f"$c $tag: {d['value']:.0f} Ms/s {d['ms_per_step']:.1f} ms spp/pass {d['config']['spp_per_pass']} passes {d['config']['passes_per_step']} crc {d['frame']['crc32']}")
PY
}
for rep in 1 2; do
run base_$rep c2 4
run big_$rep c2 4 --max-paths 352321536
run big3_$rep c2 4 --max-paths 528482304
done
run base c3 3
run big c3 3 --max-paths 352321536
run base c5 3
run big c5 3 --max-paths 352321536
run base c4 1
run big c4 1 --max-paths 352321536
