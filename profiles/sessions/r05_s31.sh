# round 5, session 31: co-residency knobs re-swept with dealt chunks (balanced traversal kernels)
O=gpurun_out/s31; mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline --no-extras > $O/${c}_$tag.log 2>&1 || { echo "bench failed"; tail -5 $O/${c}_$tag.log; return; }
python - <<PY
import json
d=json.loads([l for l in open('$O/${c}_$tag.log') if l.startswith('{')][-1])
print(f"$c $tag: {d['value']:.0f} Ms/s {d['ms_per_step']:.1f} ms")
PY
}
for c in c3 c5; do
run base $c 2 TUTU_TRACE_DEAL=6
for sb in 1 2 3; do run shade_bpc$sb $c 2 TUTU_TRACE_DEAL=6 TUTU_SHADE_BPC=$sb; done
for tb in 4 5 6; do run trace_bpc$tb $c 2 TUTU_TRACE_DEAL=6 TUTU_TRACE_BPC=$tb; done
run paths336 $c 2 TUTU_TRACE_DEAL=6 TUTU_PATHS_MI=336
run paths336_d0 $c 2 TUTU_TRACE_DEAL=0 TUTU_PATHS_MI=336
run sets3_paths252 $c 2 TUTU_TRACE_DEAL=6 TUTU_SETS=3 TUTU_PATHS_MI=252
done
