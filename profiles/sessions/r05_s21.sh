# round 5, session 21: the library functions of the path with the C library's bits (device_libm.h) -- parity first (per-sample radiances and
# whole frames against the reference build), then what it costs (same-box A/B against the build before)
set -x
O=gpurun_out/s21; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_hip_frames.py -m gpu -q -s -x > $O/tests.log 2>&1; echo "tests rc=$?"; grep "mean per-pixel L2\|passed\|failed\|Error\|assert" $O/tests.log | cut -c1-330 | head -40
run() { tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline > $O/${c}_$tag.log 2>&1 || { echo "bench failed"; tail -5 $O/${c}_$tag.log; return; }
python - <<PY
import json
d=json.loads([l for l in open('$O/${c}_$tag.log') if l.startswith('{')][-1]); r=d['roofline']; ex=r['exclusive_kernel_ms_per_step']
print(f"$c $tag: {d['value']:.0f} Ms/s {d['ms_per_step']:.1f} ms | exclusive: closest {ex['k_trace_closest']:.1f} any {ex['k_trace_any']:.1f} shade {ex['k_shade']:.1f} d0 {ex['k_shade_depth0']:.1f} | crc {d['frame']['crc32']} mean {d['frame']['mean']:.6f}")
PY
}
B=$PWD/build/libtutu_prelibm.so; N=$PWD/tuturenderer_amd/libtutu_hip.so
for rep in 1 2; do
run base_$rep c2 3 TUTU_HIP_LIB=$B
run libm_$rep c2 3 TUTU_HIP_LIB=$N
done
for c in c3 c5; do
run base $c 3 TUTU_HIP_LIB=$B
run libm $c 3 TUTU_HIP_LIB=$N
done
