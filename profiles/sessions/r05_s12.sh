# round 5, session 12: the leaf records of session 11 with every record in a 128-B line of its own (no straddling) -- same-box A/B
set -x
O=gpurun_out/s12; mkdir -p $O
export TMPDIR=/tmp
run() { tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline > $O/${c}_$tag.log 2>&1 || { echo "bench failed"; tail -5 $O/${c}_$tag.log; exit 1; }
python - <<PY
import json
d=json.loads([l for l in open('$O/${c}_$tag.log') if l.startswith('{')][-1]); r=d['roofline']; m=r['measured_per_ray']; t=r['traversal']; ex=r['exclusive_kernel_ms_per_step']
print(f"$c $tag: {d['value']:.0f} Ms/s | closest {ex['k_trace_closest']:.1f} any {ex['k_trace_any']:.1f} ms | N/T closest {m['N_closest']:.1f} {m['T_closest']:.1f} | crc {d['frame']['crc32']}")
PY
}
B=$PWD/tuturenderer_amd/libtutu_hip.so; N=$PWD/build/libtutu_rec128.so
for rep in 1 2; do
for c in c3 c5; do
run base_$rep $c 3 TUTU_HIP_LIB=$B
run rec128_$rep $c 3 TUTU_HIP_LIB=$N
done
done
