# round 5, session 4: the eight-wide walk with per-node visiting-order tables (kd slots) against octant slots and the four-wide tree
set -x
O=gpurun_out/s4; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_hip_wide.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/tests.log
[ $rc -ne 0 ] && exit 1
run() { tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline > $O/${c}_$tag.log 2>&1 || { echo "bench failed"; tail -5 $O/${c}_$tag.log; exit 1; }
python - <<PY
import json
d=json.loads([l for l in open('$O/${c}_$tag.log') if l.startswith('{')][-1]); r=d['roofline']; m=r['measured_per_ray']; t=r['traversal']; ex=r['exclusive_kernel_ms_per_step']
print(f"$c $tag: {d['value']:.0f} Ms/s | closest {ex['k_trace_closest']:.1f} any {ex['k_trace_any']:.1f} ms | N/T closest {m['N_closest']:.1f} {m['T_closest']:.1f} shadow {m['N_shadow']:.1f} {m['T_shadow']:.1f} | lanes node {t['lanes_active_node_step_closest']:.2f} leaf {t['lanes_active_leaf_step_closest']:.2f} any {t['lanes_active_node_step_any']:.2f} {t['lanes_active_leaf_step_any']:.2f} | crc {d['frame']['crc32']}")
PY
}
run w4 c4 1 TUTU_WIDE8=0
run w8_oct c4 1 TUTU_WIDE8_SLOTS=1
run w8_kd c4 1 TUTU_WIDE8_SLOTS=2
run w8_kd_i2 c4 1 TUTU_WIDE8_SLOTS=2 TUTU_WIDE8_INNER_STEPS=2 TUTU_WIDE8_INNER_STEPS_ANY=2
for rep in 1 2; do
run w4_$rep c3 3 TUTU_WIDE8=0
run w8_kd_$rep c3 3 TUTU_WIDE8_SLOTS=2
run w8_kd_i2_$rep c3 3 TUTU_WIDE8_SLOTS=2 TUTU_WIDE8_INNER_STEPS=2 TUTU_WIDE8_INNER_STEPS_ANY=2
run w4_$rep c5 3 TUTU_WIDE8=0
run w8_kd_$rep c5 3 TUTU_WIDE8_SLOTS=2
run w8_kd_i2_$rep c5 3 TUTU_WIDE8_SLOTS=2 TUTU_WIDE8_INNER_STEPS=2 TUTU_WIDE8_INNER_STEPS_ANY=2
done
