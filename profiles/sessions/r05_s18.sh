# round 5, session 18: the eight-wide walk's round shape re-swept on c5 and c3 (final tree)
O=gpurun_out/s18; mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; c=$2; shift 2
env "$@" timeout -k 10 300 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/${c}_$tag.log 2>&1 || { echo "bench failed $c $tag"; return; }
python - <<PY
import json
d=json.loads([l for l in open('$O/${c}_$tag.log') if l.startswith('{')][-1]); print(f"$c $tag: {d['value']:.0f} Ms/s {d['ms_per_step']:.1f} ms")
PY
}
for c in c5 c3; do
run base1 $c TUTU_X=1
run i2a1 $c TUTU_WIDE8_INNER_STEPS_ANY=1
run i2a3 $c TUTU_WIDE8_INNER_STEPS_ANY=3
run i3a2 $c TUTU_WIDE8_INNER_STEPS=3
run i1a2 $c TUTU_WIDE8_INNER_STEPS=1
run l1 $c TUTU_WIDE8_LEAF_STEPS=1
run l3 $c TUTU_WIDE8_LEAF_STEPS=3
run la8 $c TUTU_WIDE8_LEAF_AGAIN=8
run la32 $c TUTU_WIDE8_LEAF_AGAIN=32
run rf16 $c TUTU_REFILL_MIN=16
run rf32 $c TUTU_REFILL_MIN=32
run room10 $c TUTU_WIDE8_LEAF_ROOM=10
run base2 $c TUTU_X=1
done
