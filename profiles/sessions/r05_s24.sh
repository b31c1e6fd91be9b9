# round 5, session 24: knob exact_sum -- the radiance folded from the deepest vertex back, as the reference's recursion returns it.
# Are the per-sample radiances and the whole frames then the reference build's, bit for bit?  And what does it cost?
set -x
O=gpurun_out/s24; mkdir -p $O
export TMPDIR=/tmp
TUTU_EXACT_SUM=1 timeout -k 10 300 python tests/tools/bit_census.py > $O/census.log 2>&1; echo "census rc=$?"; grep "^samples" $O/census.log
TUTU_EXACT_SUM=1 timeout -k 10 600 python - > $O/frames.log 2>&1 <<'PY'
import sys, zlib
import numpy as np
sys.path.insert(0, "tests")
from conftest import golden_path
import tuturenderer_amd as tr
from tuturenderer_amd import scenes
tr.load_library()
for name, mk in (("c1", lambda: scenes.cornell_box(800, 800)), ("c2", lambda: scenes.cornell_box(800, 800)), ("c5", lambda: scenes.veach_room(800, 600, small_light=False)),
                 ("c3", lambda: scenes.bunny_box(1024, 1024)), ("c4", lambda: scenes.broom_room(1600, 900)), ("c4_64", lambda: scenes.broom_room(1600, 900))):
    z = np.load(golden_path(f"frame_{name}.npz"))
    with tr.Context(mk()) as ctx:
        f = ctx.render(int(z["spp"]), int(z["key0"]), int(z["key1"]))
    w = z["rgb"]
    diff = (f.view(np.uint32) != w.view(np.uint32)).any(-1)
    print(f"frame {name}: pixels whose bits differ from the reference build's {int(diff.sum())} of {diff.size}; crc {zlib.crc32(f.tobytes()):08x} / reference {zlib.crc32(np.ascontiguousarray(w).tobytes()):08x}", flush=True)
PY
echo "frames rc=$?"; cat $O/frames.log | tail -8
run() { tag=$1; c=$2; st=$3; shift 3
env "$@" timeout -k 10 300 python bench.py --config $c --steps $st --warmup 1 --no-cpu-baseline > $O/${c}_$tag.log 2>&1 || { echo "bench failed"; tail -5 $O/${c}_$tag.log; return; }
python - <<PY
import json
d=json.loads([l for l in open('$O/${c}_$tag.log') if l.startswith('{')][-1]); r=d['roofline']; ex=r['exclusive_kernel_ms_per_step']
print(f"$c $tag: {d['value']:.0f} Ms/s {d['ms_per_step']:.1f} ms | exclusive: closest {ex['k_trace_closest']:.1f} any {ex['k_trace_any']:.1f} shade {ex['k_shade']:.1f} d0 {ex['k_shade_depth0']:.1f} | crc {d['frame']['crc32']} mean {d['frame']['mean']:.6f}")
PY
}
for c in c2 c3 c5 c4; do
run fwd $c 3 TUTU_EXACT_SUM=0
run xsum $c 3 TUTU_EXACT_SUM=1
done
