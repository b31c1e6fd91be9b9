# round 5, session 32: the top of the eight-wide tree staged in LDS (knob wide8_top = node ids staged at most)
O=gpurun_out/s32; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_wide.py -m gpu -q -x > $O/wide.log 2>&1; echo "wide rc=$?"; tail -2 $O/wide.log
run() { tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline > $O/${c}_$tag.log 2>&1 || { echo "bench failed"; tail -5 $O/${c}_$tag.log; return; }
python - <<PY
import json
d=json.loads([l for l in open('$O/${c}_$tag.log') if l.startswith('{')][-1]); r=d['roofline']; ex=r['exclusive_kernel_ms_per_step']
print(f"$c $tag: {d['value']:.0f} Ms/s {d['ms_per_step']:.1f} ms | exclusive: closest {ex['k_trace_closest']:.1f} any {ex['k_trace_any']:.1f} shade {ex['k_shade']:.1f} d0 {ex['k_shade_depth0']:.1f} | crc {d['frame']['crc32']}")
PY
}
for c in c3 c5; do
for t in 0 16 40 80; do
run top$t $c 3 TUTU_WIDE8_TOP=$t
done
done
run w8_top0 c4 2 TUTU_WIDE8=2 TUTU_WIDE8_TOP=0
run w8_top80 c4 2 TUTU_WIDE8=2 TUTU_WIDE8_TOP=80
run w4 c4 2
