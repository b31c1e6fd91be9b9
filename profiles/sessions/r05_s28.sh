# round 5, session 28: trace_deal 0 vs 6, overlapped frames, alternating repeats (is c4's -1 % real?)
O=gpurun_out/s28; mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline --no-extras > $O/${c}_$tag.log 2>&1 || { echo "bench failed"; tail -5 $O/${c}_$tag.log; return; }
python - <<PY
import json
d=json.loads([l for l in open('$O/${c}_$tag.log') if l.startswith('{')][-1])
print(f"$c $tag: {d['value']:.0f} Ms/s {d['ms_per_step']:.1f} ms")
PY
}
for rep in 1 2 3; do
for c in c4 c3 c5; do
run deal0_$rep $c 3 TUTU_TRACE_DEAL=0
run deal6_$rep $c 3 TUTU_TRACE_DEAL=6
done
done
