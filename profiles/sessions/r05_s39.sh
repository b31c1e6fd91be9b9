# round 5, session 39: powf's tables in LDS, staged only when the scene needs them -- the whole suite, then c2 against the build before
O=gpurun_out/s39; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/gpu_tests.log
run() { tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 200 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline --no-extras > $O/${c}_$tag.log 2>&1 || { echo "bench failed"; tail -5 $O/${c}_$tag.log; return; }
python - <<PY
import json
d=json.loads([l for l in open('$O/${c}_$tag.log') if l.startswith('{')][-1])
print(f"$c $tag: {d['value']:.0f} Ms/s {d['ms_per_step']:.1f} ms | crc {d['frame']['crc32']}")
PY
}
for rep in 1 2 3; do
run base_$rep c2 3 TUTU_HIP_LIB=$PWD/build/libtutu_base.so
run lds_$rep c2 3 TUTU_HIP_LIB=$PWD/tuturenderer_amd/libtutu_hip.so
done
