# round 5, session 40: exact_sum with a level's two 16-B halves adjacent (one 32-B piece per level and path) -- frames still the reference's; cost
O=gpurun_out/s40; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_frames.py tests/test_hip_parity.py -m gpu -q -x -k "exact_sum" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
run() { tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 200 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline > $O/${c}_$tag.log 2>&1 || { echo "bench failed"; tail -5 $O/${c}_$tag.log; return; }
python - <<PY
import json
d=json.loads([l for l in open('$O/${c}_$tag.log') if l.startswith('{')][-1]); r=d['roofline']; ex=r['exclusive_kernel_ms_per_step']
print(f"$c $tag: {d['value']:.0f} Ms/s {d['ms_per_step']:.1f} ms | exclusive: closest {ex['k_trace_closest']:.1f} any {ex['k_trace_any']:.1f} shade {ex['k_shade']:.1f} d0 {ex['k_shade_depth0']:.1f} other {ex.get('other', 0):.1f} | crc {d['frame']['crc32']}")
PY
}
for c in c2 c3 c5; do
run fwd $c 3 TUTU_EXACT_SUM=0
run xsum $c 3 TUTU_EXACT_SUM=1
done
