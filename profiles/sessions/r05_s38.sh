# round 5, session 38: powf's tables in LDS (device_math.h: g_pow_lds) -- parity (every kernel that reaches pow5f stages them), then cost
O=gpurun_out/s38; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_hip_integrators.py tests/test_hip_frames.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
run() { tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 200 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline > $O/${c}_$tag.log 2>&1 || { echo "bench failed"; tail -5 $O/${c}_$tag.log; return; }
python - <<PY
import json
d=json.loads([l for l in open('$O/${c}_$tag.log') if l.startswith('{')][-1]); r=d['roofline']; ex=r['exclusive_kernel_ms_per_step']
print(f"$c $tag: {d['value']:.0f} Ms/s {d['ms_per_step']:.1f} ms | exclusive: closest {ex['k_trace_closest']:.1f} any {ex['k_trace_any']:.1f} shade {ex['k_shade']:.1f} d0 {ex['k_shade_depth0']:.1f} | crc {d['frame']['crc32']}")
PY
}
for c in c3 c5 c2; do
run base_1 $c 3 TUTU_HIP_LIB=$PWD/build/libtutu_base.so
run lds_1 $c 3 TUTU_HIP_LIB=$PWD/tuturenderer_amd/libtutu_hip.so
run base_2 $c 3 TUTU_HIP_LIB=$PWD/build/libtutu_base.so
run lds_2 $c 3 TUTU_HIP_LIB=$PWD/tuturenderer_amd/libtutu_hip.so
done
