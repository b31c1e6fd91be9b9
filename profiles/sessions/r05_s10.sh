# round 5, session 10: c2 -- how many blocks of the shade / flat-scan kernels should share a CU while four passes overlap (re-swept with
# the round-4 flat kernels; the earlier sweeps were made with the tree-walk kernels)
O=gpurun_out/s10; mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; shift
env "$@" timeout -k 10 300 python bench.py --config c2 --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $O/c2_$tag.log 2>&1 || { echo "bench failed $tag"; tail -3 $O/c2_$tag.log; return; }
python - <<PY
import json
d=json.loads([l for l in open('$O/c2_$tag.log') if l.startswith('{')][-1]); print(f"c2 $tag: {d['value']:.0f} Ms/s {d['ms_per_step']:.1f} ms crc {d['frame']['crc32']}")
PY
}
for rep in 1 2; do
run base_$rep TUTU_X=1
run sbpc2_$rep TUTU_SHADE_BPC=2
run sbpc3_$rep TUTU_SHADE_BPC=3
run tbpc4_$rep TUTU_TRACE_BPC=4
run tbpc3_s2_$rep TUTU_TRACE_BPC=3 TUTU_SHADE_BPC=2
run tbpc6_$rep TUTU_TRACE_BPC=6
run sets3_$rep TUTU_SETS=3
run sets2_$rep TUTU_SETS=2
done
