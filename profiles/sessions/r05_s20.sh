# round 5, session 20: the passes of a frame's last round tapered (streams start staggered, so they end staggered: 10-13 ms of a c2 frame with
# one or two streams left) -- TUTU_PASS_TAPER in per mille
O=gpurun_out/s20; mkdir -p $O
export TMPDIR=/tmp
export TUTU_HIP_LIB=$PWD/build/libtutu_taper.so
run() { tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 2 --no-cpu-baseline --no-extras > $O/${c}_$tag.log 2>&1 || { echo "bench failed $c $tag"; tail -3 $O/${c}_$tag.log; return; }
python - <<PY
import json
d=json.loads([l for l in open('$O/${c}_$tag.log') if l.startswith('{')][-1]); print(f"$c $tag: {d['value']:.0f} Ms/s {d['ms_per_step']:.2f} ms crc {d['frame']['crc32']}")
PY
}
for rep in 1 2; do
for t in 0 50 100 200 300 450; do run t${t}_$rep c2 5 TUTU_PASS_TAPER=$t; done
done
for t in 0 100 200 300; do run t$t c3 3 TUTU_PASS_TAPER=$t; done
for t in 0 100 200 300; do run t$t c5 3 TUTU_PASS_TAPER=$t; done
