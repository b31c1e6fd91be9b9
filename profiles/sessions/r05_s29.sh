# round 5, session 29: passes in flight x dealing (do balanced kernels want less overlap?)
O=gpurun_out/s29; mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline --no-extras > $O/${c}_$tag.log 2>&1 || { echo "bench failed"; tail -5 $O/${c}_$tag.log; return; }
python - <<PY
import json
d=json.loads([l for l in open('$O/${c}_$tag.log') if l.startswith('{')][-1])
print(f"$c $tag: {d['value']:.0f} Ms/s {d['ms_per_step']:.1f} ms")
PY
}
for c in c3 c5 c4; do
for sets in 1 2 3 4; do
for dl in 0 6; do
run sets${sets}_deal$dl $c 2 TUTU_SETS=$sets TUTU_TRACE_DEAL=$dl
done
done
done
