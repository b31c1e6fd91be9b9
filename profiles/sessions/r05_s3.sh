# round 5, session 3: what the eight-wide walk is sensitive to -- slot order, round shape, blocks per CU (c4 and c3, one box)
set -x
O=gpurun_out/s3; mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline > $O/${c}_$tag.log 2>&1 || { echo "bench failed"; tail -5 $O/${c}_$tag.log; exit 1; }
python - <<PY
import json
d=json.loads([l for l in open('$O/${c}_$tag.log') if l.startswith('{')][-1]); r=d['roofline']; m=r['measured_per_ray']; t=r['traversal']; ex=r['exclusive_kernel_ms_per_step']
print(f"$c $tag: {d['value']:.0f} Ms/s | closest {ex['k_trace_closest']:.1f} any {ex['k_trace_any']:.1f} ms | N/T closest {m['N_closest']:.1f} {m['T_closest']:.1f} shadow {m['N_shadow']:.1f} {m['T_shadow']:.1f} | lanes node {t['lanes_active_node_step_closest']:.2f} leaf {t['lanes_active_leaf_step_closest']:.2f} any {t['lanes_active_node_step_any']:.2f} {t['lanes_active_leaf_step_any']:.2f} | crc {d['frame']['crc32']}")
PY
}
run w4 c4 1 TUTU_WIDE8=0
run w8 c4 1 TUTU_WIDE8=1
run w8_slots0 c4 1 TUTU_WIDE8_SLOTS=0
run w8_i1 c4 1 TUTU_WIDE8_INNER_STEPS=1 TUTU_WIDE8_INNER_STEPS_ANY=1
run w8_i2 c4 1 TUTU_WIDE8_INNER_STEPS=2 TUTU_WIDE8_INNER_STEPS_ANY=2
run w8_i4 c4 1 TUTU_WIDE8_INNER_STEPS=4 TUTU_WIDE8_INNER_STEPS_ANY=4
run w8_i2l3 c4 1 TUTU_WIDE8_INNER_STEPS=2 TUTU_WIDE8_INNER_STEPS_ANY=2 TUTU_WIDE8_LEAF_STEPS=3 TUTU_WIDE8_LEAF_AGAIN=8
run w8_i1l2a1 c4 1 TUTU_WIDE8_INNER_STEPS=1 TUTU_WIDE8_INNER_STEPS_ANY=1 TUTU_WIDE8_LEAF_STEPS=2 TUTU_WIDE8_LEAF_AGAIN=1
run w8_bpc6 c4 1 TUTU_TRACE_BPC=6
run w4 c3 3 TUTU_WIDE8=0
run w8 c3 3 TUTU_WIDE8=1
run w8_slots0 c3 3 TUTU_WIDE8_SLOTS=0
run w8_i1 c3 3 TUTU_WIDE8_INNER_STEPS=1 TUTU_WIDE8_INNER_STEPS_ANY=1
run w8_i2 c3 3 TUTU_WIDE8_INNER_STEPS=2 TUTU_WIDE8_INNER_STEPS_ANY=2
run w8_i4 c3 3 TUTU_WIDE8_INNER_STEPS=4 TUTU_WIDE8_INNER_STEPS_ANY=4
run w8_noearly c3 3 TUTU_WIDE_EARLY=0
run w8_slots0 c5 3 TUTU_WIDE8_SLOTS=0
run w8_i2 c5 3 TUTU_WIDE8_INNER_STEPS=2 TUTU_WIDE8_INNER_STEPS_ANY=2
