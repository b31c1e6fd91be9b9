#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- whole frames of the BASELINE configs rendered by the REFERENCE'S OWN CODE.

Run in the build container (needs /root/reference and oracle/_ref/libtutu_ref.so):

    make -C oracle && python oracle/gen_frames.py [c2 c5 c3 c4 native]

    c2  Cornell box 800x800, 512 spp      (BASELINE configs[1], the headline)          -> tests/golden/frame_c2.npz
    c5  veach room 800x600, 512 spp       (configs[4], PathTracing)                    -> tests/golden/frame_c5.npz
    c3  bunny stand-in 1024x1024, 256 spp (configs[2])                                 -> tests/golden/frame_c3.npz
    c4  broom stand-in 1600x900, 16 spp   (configs[3] at a reduced spp: the reference traces 0.05 Msamples/s here)
                                                                                        -> tests/golden/frame_c4.npz
    c1  Cornell box 800x800, 16 spp       (configs[0], the reference's own CPU-runnable case)   -> tests/golden/frame_c1.npz
    c4_64  broom stand-in 1600x900, 64 spp (configs[3] at 64 of its 1024 spp, ~1 h of CPU)       -> tests/golden/frame_c4_64.npz
    native  Cornell box 128x128, 4096 spp with the reference's OWN std::mt19937 (no engine swap),
            from oracle/_ref/libtutu_ref_native.so                                      -> tests/golden/frame_native_cornell.npz

What a file holds: `rgb` = the float32 frame buffer PathTracing's sub_render_pt leaves in g->cam.FrameBuffer.rgb
(PathTracing.hpp:485-516: per pixel, SPP calls of the reference's own traceRay on the pixel's ray, NaN samples dropped,
x 1/SPP) -- produced by oracle/ref_harness.cpp: tor_render, which calls the reference's traceRay compiled from
/root/reference/include where it lies -- plus the Philox key, spp and a checksum of the scene arrays.  Matched seed: the
reference consumes stream (pixel, sample) of Philox key (KEY0, key1) through the engine swap of oracle/ref_shim.h, the same
stream the port and the HIP kernels draw from.  Data only; no reference source text is stored.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import parity_cases as pc  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402
from tuturenderer_amd import scenes  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

# the same scenes, keys and spp as bench.py's configs() -- c4 at a reduced spp
FRAMES = {
    "c2": dict(mk=lambda: scenes.cornell_box(800, 800), key1=2, spp=512),
    "c5": dict(mk=lambda: scenes.veach_room(800, 600, small_light=False), key1=5, spp=512),
    "c3": dict(mk=lambda: scenes.bunny_box(1024, 1024), key1=3, spp=256),
    "c4": dict(mk=lambda: scenes.broom_room(1600, 900), key1=4, spp=16),
    # round 5: the two thin pins.  c1 = BASELINE configs[0] whole; c4_64 = the broom frame at four times the spp of "c4" (same key,
    # so its samples 0..15 are c4's): flip noise of single samples falls as 1/spp, a bias would not
    "c1": dict(mk=lambda: scenes.cornell_box(800, 800), key1=1, spp=16),
    "c4_64": dict(mk=lambda: scenes.broom_room(1600, 900), key1=4, spp=64),
}
NATIVE = dict(mk=lambda: scenes.cornell_box(128, 128), spp=4096)


def _ggx_mirror():
    from oracle.pyoracle import MICROFACET_T, PERFECT_REFLECTIVE, make_material

    return scenes.cornell_box(128, 128, tall=make_material(MICROFACET_T, (0.725, 0.71, 0.68), eta=1.5, roughness=0.2),
                              short=make_material(PERFECT_REFLECTIVE, (0.725, 0.71, 0.68)))


# more native-RNG frames (SURVEY.md 8c's list): name -> (scene, spp); written to tests/golden/frame_<name>.npz
NATIVE_MORE = {
    "native_veach": dict(mk=lambda: scenes.veach_room(160, 120, small_light=False), spp=2048),
    "native_ggxT_mirror": dict(mk=_ggx_mirror, spp=2048),
}


def scene_crc(sc):
    return pc.checksum(np.ascontiguousarray(sc["verts"], np.float32), np.ascontiguousarray(sc["normals"], np.float32),
                       np.ascontiguousarray(sc["mat_id"], np.int32))


def main(which):
    os.makedirs(GOLD, exist_ok=True)
    R = None
    for name in which:
        if name == "native" or name in NATIVE_MORE:
            RN = Oracle("reference_native")
            cfgn = NATIVE if name == "native" else NATIVE_MORE[name]
            sc = cfgn["mk"]()
            S = RN.scene(sc)
            t0 = time.time()
            rgb = S.render(cfgn["spp"], 0, 0)  # the key is not consumed: the reference's thread_local mt19937 draws
            dt = time.time() - t0
            S.close()
            out = "frame_native_cornell.npz" if name == "native" else f"frame_{name}.npz"
            np.savez_compressed(os.path.join(GOLD, out), rgb=rgb, spp=np.int32(cfgn["spp"]), scene_crc=scene_crc(sc))
            print(f"{name}: {sc['width']}x{sc['height']}x{cfgn['spp']} in {dt:.1f} s, mean {rgb.mean():.5f}", flush=True)
            continue
        if R is None:
            R = Oracle("reference")
        cfg = FRAMES[name]
        sc = cfg["mk"]()
        S = R.scene(sc)
        t0 = time.time()
        rgb = S.render(cfg["spp"], pc.KEY0, cfg["key1"], nthreads=int(os.environ.get("TUTU_GEN_THREADS", "0")) or None)
        dt = time.time() - t0
        S.close()
        n = sc["width"] * sc["height"] * cfg["spp"]
        np.savez_compressed(os.path.join(GOLD, f"frame_{name}.npz"), rgb=rgb, spp=np.int32(cfg["spp"]), key0=np.uint32(pc.KEY0),
                            key1=np.uint32(cfg["key1"]), scene_crc=scene_crc(sc))
        print(f"{name}: {sc['width']}x{sc['height']}x{cfg['spp']} in {dt:.1f} s ({n / dt / 1e6:.3f} Msamples/s), mean {rgb.mean():.5f}, "
              f"nan {int(np.isnan(rgb).sum())}", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or ["c2", "c5", "c3", "c4"])
