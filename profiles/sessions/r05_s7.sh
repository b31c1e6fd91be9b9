# round 5, session 7: k_shade (generic kernel of the mixed scenes) with the window's list entries and hit triangles staged in LDS: one round
# trip per chunk instead of three -- same-box A/B of two builds through TUTU_HIP_LIB, then the frame tests on the new build
set -x
O=gpurun_out/s7; mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline > $O/${c}_$tag.log 2>&1 || { echo "bench failed"; tail -5 $O/${c}_$tag.log; exit 1; }
python - <<PY
import json
d=json.loads([l for l in open('$O/${c}_$tag.log') if l.startswith('{')][-1]); r=d['roofline']; ex=r['exclusive_kernel_ms_per_step']; ks=r['per_kernel']['k_shade']
print(f"$c $tag: {d['value']:.0f} Ms/s {d['ms_per_step']:.1f} ms | exclusive: closest {ex['k_trace_closest']:.1f} any {ex['k_trace_any']:.1f} shade {ex['k_shade']:.1f} d0 {ex['k_shade_depth0']:.1f} | k_shade frac {ks['frac']:.3f} launch {ks['avg_launch_ms']:.3f} ms | crc {d['frame']['crc32']}")
PY
}
B=$PWD/tuturenderer_amd/libtutu_hip.so; N=$PWD/build/libtutu_sh1.so
for rep in 1 2; do
for c in c3 c5; do
run base_$rep $c 3 TUTU_HIP_LIB=$B
run new_$rep $c 3 TUTU_HIP_LIB=$N
done
done
run base c4 1 TUTU_HIP_LIB=$B
run new c4 1 TUTU_HIP_LIB=$N
run base c2 3 TUTU_HIP_LIB=$B
run new c2 3 TUTU_HIP_LIB=$N
