# round 5, session 30: what each restated library function costs the shade kernels -- measuring builds with ONE function replaced by the
# device library's (build/libtutu_cost_*.so), exclusive kernel times of c3 / c5 / c2
O=gpurun_out/s30; mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline > $O/${c}_$tag.log 2>&1 || { echo "bench failed"; tail -5 $O/${c}_$tag.log; return; }
python - <<PY
import json
d=json.loads([l for l in open('$O/${c}_$tag.log') if l.startswith('{')][-1]); r=d['roofline']; ex=r['exclusive_kernel_ms_per_step']
print(f"$c $tag: {d['value']:.0f} Ms/s {d['ms_per_step']:.1f} ms | exclusive: closest {ex['k_trace_closest']:.1f} any {ex['k_trace_any']:.1f} shade {ex['k_shade']:.1f} d0 {ex['k_shade_depth0']:.1f}")
PY
}
for c in c3 c5 c2; do
run all $c 3 TUTU_HIP_LIB=$PWD/tuturenderer_amd/libtutu_hip.so
for v in POW SINCOS ACOS TAN; do
run fast_$v $c 3 TUTU_HIP_LIB=$PWD/build/libtutu_cost_$v.so
done
run none $c 3 TUTU_HIP_LIB=$PWD/build/libtutu_prelibm.so
done
