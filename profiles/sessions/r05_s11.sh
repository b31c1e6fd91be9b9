# round 5, session 11: the eight-wide walk with ONE 80-B record per leaf (triangle + reference's leaf box + reference, in the tree's own
# leaf order: five requests and no indirection per leaf test instead of six) -- forms test, then same-box A/B against the build before
set -x
O=gpurun_out/s11; mkdir -p $O
export TMPDIR=/tmp
rc=0

run() { tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline > $O/${c}_$tag.log 2>&1 || { echo "bench failed"; tail -5 $O/${c}_$tag.log; exit 1; }
python - <<PY
import json
d=json.loads([l for l in open('$O/${c}_$tag.log') if l.startswith('{')][-1]); r=d['roofline']; m=r['measured_per_ray']; t=r['traversal']; ex=r['exclusive_kernel_ms_per_step']
print(f"$c $tag: {d['value']:.0f} Ms/s | closest {ex['k_trace_closest']:.1f} any {ex['k_trace_any']:.1f} ms | N/T closest {m['N_closest']:.1f} {m['T_closest']:.1f} shadow {m['N_shadow']:.1f} {m['T_shadow']:.1f} | lanes node {t['lanes_active_node_step_closest']:.2f} leaf {t['lanes_active_leaf_step_closest']:.2f} | crc {d['frame']['crc32']}")
PY
}
B=$PWD/build/libtutu_sh1.so; N=$PWD/tuturenderer_amd/libtutu_hip.so
for rep in 1 2; do
for c in c3 c5; do
run base_$rep $c 3 TUTU_HIP_LIB=$B
run rec_$rep $c 3 TUTU_HIP_LIB=$N
done
done
run base_w8 c4 1 TUTU_HIP_LIB=$B TUTU_WIDE8=2
run rec_w8 c4 1 TUTU_HIP_LIB=$N TUTU_WIDE8=2
