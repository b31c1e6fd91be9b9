# round 5, session 36: the rays of every stage traced grouped by a 15-bit Morton code of their origin (knob ray_sort)
O=gpurun_out/s36; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_wide.py -m gpu -q -x > $O/wide.log 2>&1; echo "wide rc=$?"; tail -2 $O/wide.log
run() { tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline > $O/${c}_$tag.log 2>&1 || { echo "bench failed"; tail -5 $O/${c}_$tag.log; return; }
python - <<PY
import json
d=json.loads([l for l in open('$O/${c}_$tag.log') if l.startswith('{')][-1]); r=d['roofline']; ex=r['exclusive_kernel_ms_per_step']
print(f"$c $tag: {d['value']:.0f} Ms/s {d['ms_per_step']:.1f} ms | exclusive: closest {ex['k_trace_closest']:.1f} any {ex['k_trace_any']:.1f} shade {ex['k_shade']:.1f} d0 {ex['k_shade_depth0']:.1f} other {ex.get('other', 0):.1f} | crc {d['frame']['crc32']}")
PY
}
run sort0 c4 2 TUTU_RAY_SORT=0
run sort1 c4 2 TUTU_RAY_SORT=1
run sort0_b c4 2 TUTU_RAY_SORT=0
run sort1_b c4 2 TUTU_RAY_SORT=1
for c in c3 c5; do
run sort0 $c 3 TUTU_RAY_SORT=0
run sort1 $c 3 TUTU_RAY_SORT=1
done
