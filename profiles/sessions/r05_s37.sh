# round 5, session 37: ray_sort with wave-aggregated atomics -- first a SMALL frame (bounded), then the frames
O=gpurun_out/s37; mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; c=$2; st=$3; lim=$4; shift 4
env "$@" timeout -k 10 $lim python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline > $O/${c}_$tag.log 2>&1 || { echo "bench failed $c $tag"; tail -3 $O/${c}_$tag.log; return 1; }
python - <<PY
import json
d=json.loads([l for l in open('$O/${c}_$tag.log') if l.startswith('{')][-1]); r=d['roofline']; ex=r['exclusive_kernel_ms_per_step']
print(f"$c $tag: {d['value']:.0f} Ms/s {d['ms_per_step']:.1f} ms | exclusive: closest {ex['k_trace_closest']:.1f} any {ex['k_trace_any']:.1f} shade {ex['k_shade']:.1f} d0 {ex['k_shade_depth0']:.1f} other {ex.get('other', 0):.1f} | crc {d['frame']['crc32']}")
PY
}
# (bench.py --spp: a 58-spp frame = two passes)
small() { tag=$1; shift
env "$@" timeout -k 10 90 python bench.py --config c4 --spp 58 --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $O/c4_small_$tag.log 2>&1 || { echo "small failed $tag"; tail -3 $O/c4_small_$tag.log; return 1; }
python - <<PY
import json
d=json.loads([l for l in open('$O/c4_small_$tag.log') if l.startswith('{')][-1])
print(f"c4 58 spp $tag: {d['value']:.0f} Ms/s {d['ms_per_step']:.1f} ms crc {d['frame']['crc32']}")
PY
}
small sort0 TUTU_RAY_SORT=0 && small sort1 TUTU_RAY_SORT=1 || exit 1
run sort0 c4 2 120 TUTU_RAY_SORT=0 && run sort1 c4 2 120 TUTU_RAY_SORT=1 && run sort0_b c4 2 120 TUTU_RAY_SORT=0 && run sort1_b c4 2 120 TUTU_RAY_SORT=1
run sort0 c3 2 60 TUTU_RAY_SORT=0 && run sort1 c3 2 60 TUTU_RAY_SORT=1
