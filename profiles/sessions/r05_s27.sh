# round 5, session 27: the persistent walks take their rays in chunks dealt round-robin (knob trace_deal = log2 of the chunk) instead of
# one contiguous range per wave.  Hits first (the wide-tree tests), then frames.
set -x
O=gpurun_out/s27; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_wide.py -m gpu -q -x > $O/wide.log 2>&1; echo "wide rc=$?"; tail -2 $O/wide.log
run() { tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline > $O/${c}_$tag.log 2>&1 || { echo "bench failed"; tail -5 $O/${c}_$tag.log; return; }
python - <<PY
import json
d=json.loads([l for l in open('$O/${c}_$tag.log') if l.startswith('{')][-1]); r=d['roofline']; ex=r['exclusive_kernel_ms_per_step']
print(f"$c $tag: {d['value']:.0f} Ms/s {d['ms_per_step']:.1f} ms | exclusive: closest {ex['k_trace_closest']:.1f} any {ex['k_trace_any']:.1f} shade {ex['k_shade']:.1f} d0 {ex['k_shade_depth0']:.1f} | crc {d['frame']['crc32']} mean {d['frame']['mean']:.6f}")
PY
}
for c in c3 c5 c4; do
for dl in 0 6 8 10 12; do
run deal$dl $c 3 TUTU_TRACE_DEAL=$dl
done
done
run deal0 c2 3 TUTU_TRACE_DEAL=0
run deal6 c2 3 TUTU_TRACE_DEAL=6
