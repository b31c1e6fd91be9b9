#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- writes the golden vectors under tests/golden/ and the veach scene dump.

Run in the build container (needs /root/reference and oracle/_ref/libtutu_ref.so):

    make -C oracle && python oracle/gen_golden.py

Every output stored here was produced by the REFERENCE'S OWN CODE (oracle/ref_harness.cpp compiled against
/root/reference/include), on inputs regenerated from fixed seeds by oracle/parity_cases.py.  The files hold data
only (arrays of inputs' checksums and expected outputs); nothing of the reference's source text is stored.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import parity_cases as pc  # noqa: E402
from oracle.pyoracle import (MICROFACET_R, MICROFACET_T, PERFECT_REFLECTIVE, PERFECT_REFRACTIVE, Oracle,  # noqa: E402
                             make_material)
from tuturenderer_amd import scenes  # noqa: E402

REF_MODEL = "/root/reference/model"
GOLD = os.path.join(ROOT, "tests", "golden")


def golden_scenes():
    """name -> (scene dict, Philox key word 1).  Small images: the vectors are per-sample, not per-image."""
    return {
        "cornell": (lambda: scenes.cornell_box(96, 96), 1),
        "cornell_ggxT_mirror": (lambda: scenes.cornell_box(
            96, 96, tall=make_material(MICROFACET_T, (0.725, 0.71, 0.68), eta=1.5, roughness=0.2),
            short=make_material(PERFECT_REFLECTIVE, (0.725, 0.71, 0.68))), 2),
        "cornell_ggxR_glass": (lambda: scenes.cornell_box(
            96, 96, tall=make_material(MICROFACET_R, (0.725, 0.71, 0.68), roughness=0.3, metallic=0.5),
            short=make_material(PERFECT_REFRACTIVE, (0.725, 0.71, 0.68), eta=1.5)), 3),
        "veach": (lambda: scenes.veach_room(96, 72, small_light=False), 5),
        "veach_slight": (lambda: scenes.veach_room(96, 72, small_light=True), 6),
        "cornell_degenerate": (lambda: pc.degenerate_cornell(64, 64), 9),
        "cornell_textured": (lambda: scenes.cornell_textured(96, 96), 11),
        "cornell_spheres": (lambda: scenes.cornell_spheres(96, 96), 12),
    }


def main():
    R = Oracle("reference")
    os.makedirs(GOLD, exist_ok=True)

    # ---- scene data: the reference's own OBJ loading of its model files
    obj = {}
    for name, v in scenes.cornell_parts():
        rv, rn = R.ref_load_obj(f"{REF_MODEL}/cornellBox/{name}.obj")
        obj[f"cornell_{name}_verts"] = rv
        obj[f"cornell_{name}_normals"] = rn
    np.savez_compressed(os.path.join(GOLD, "obj_cornell.npz"), **obj)
    veach = {}
    for name in ["room", "Llight", "sLight", "table", "glass", "tallLamp", "wallLamp"]:
        rv, rn = R.ref_load_obj(f"{REF_MODEL}/veach_bdpt/veach_{name}.obj")
        veach[f"{name}_verts"] = rv
        veach[f"{name}_normals"] = rn
        print("veach", name, len(rv))
    os.makedirs(scenes.DATA, exist_ok=True)
    np.savez_compressed(os.path.join(scenes.DATA, "veach_room.npz"), **veach)

    # ---- function-level vectors
    fn = {}
    for k, v in pc.run_bbox(R).items():
        fn[f"bbox.{k}"] = v
    for k, v in pc.run_tri(R).items():
        fn[f"tri.{k}"] = v
    for k, v in pc.run_math(R).items():
        fn[f"math.{k}"] = v
    for name, m in pc.material_set():
        fn.update(pc.run_material(R, name, m))
    # Philox known answers (Random123 kat_vectors) + the xi stream of one sample
    fn["philox.ctr"] = np.array([[0, 0, 0, 0], [0xFFFFFFFF] * 4, [0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344]], np.uint32)
    fn["philox.key"] = np.array([[0, 0], [0xFFFFFFFF, 0xFFFFFFFF], [0xA4093822, 0x299F31D0]], np.uint32)
    fn["philox.out"] = np.stack([R.philox(fn["philox.ctr"][i:i + 1], *[int(x) for x in fn["philox.key"][i]])[0] for i in range(3)])
    fn["philox.stream"] = R.rng_stream(12345, 7, pc.KEY0, 2, 64)
    for k, v in pc.run_texture(R, scenes.procedural_maps()["diffuse"][0]).items():
        fn[f"texture.{k}"] = v
    for k, v in pc.run_postprocess(R).items():
        fn[f"post.{k}"] = v
    np.savez_compressed(os.path.join(GOLD, "functions.npz"), **fn)

    # ---- scene-level vectors + per-sample radiance
    for name, (mk, key1) in golden_scenes().items():
        sc = mk()
        S = R.scene(sc)
        out = {}
        b, leaf = S.bvh_dump()
        out["bvh.bounds"] = b
        out["bvh.leaf_tri"] = leaf
        for k, v in pc.run_scene(S).items():
            out[f"scene.{k}"] = v
        for k, v in pc.run_samples(S, key1).items():
            out[f"samples.{k}"] = v
        # a small full render (spp 16) for the image-level check
        out["render.rgb"] = S.render(16, pc.KEY0, key1)
        np.savez_compressed(os.path.join(GOLD, f"scene_{name}.npz"), **out)
        L = out["samples.L"]
        print(name, "tris", len(sc["verts"]), "nodes", len(leaf), "mean L", float(np.nanmean(L)), "nan", int(np.isnan(L).sum()),
              "closest/sample", float(out["samples.nclosest"].mean()), "img mean", float(out["render.rgb"].mean()))
        S.close()


if __name__ == "__main__" and len(sys.argv) == 1:
    main()


# ---- BASELINE configs 3 and 4: the synthetic bunny / broom stand-ins.  The scenes are procedural
# (tuturenderer_amd/scenes.py regenerates them from fixed seeds), so only sample ids and the REFERENCE BUILD's answers are
# stored: per-sample radiance from its own traceRay, and its closest hits for the samples' primary rays.
def standin_scenes():
    return {
        "bunny": (lambda: scenes.bunny_box(256, 256), 3),
        "broom": (lambda: scenes.broom_room(320, 180), 4),
    }


def standin_sample_ids(sc, n=4000, seed=5):
    rng = np.random.default_rng(seed)
    pix = rng.integers(0, sc["width"] * sc["height"], n).astype(np.uint32)
    smp = rng.integers(0, 64, n).astype(np.uint32)
    return pix, smp


def main_standins():
    R = Oracle("reference")
    for name, (mk, key1) in standin_scenes().items():
        sc = mk()
        S = R.scene(sc)
        pix, smp = standin_sample_ids(sc)
        L, nd, nc = S.trace_samples(pix, smp, pc.KEY0, key1, stats=True)
        d = S.raydir((pix % sc["width"]).astype(np.int32), (pix // sc["width"]).astype(np.int32))
        o = np.repeat(S.camera()[5][None], len(pix), 0)
        hit, t, tri, *_ = S.closest(o, d)
        out = {"samples.L": L, "samples.ndraws": nd, "samples.nclosest": nc, "samples.in_crc": pc.checksum(pix, smp),
               "primary.hit": hit, "primary.t": np.where(hit.astype(bool), t, 0).astype(np.float32), "primary.tri": tri,
               "scene.crc": pc.checksum(np.ascontiguousarray(sc["verts"], np.float32), np.ascontiguousarray(sc["normals"], np.float32))}
        np.savez_compressed(os.path.join(GOLD, f"scene_{name}.npz"), **out)
        print(name, "tris", len(sc["verts"]), "mean L", float(np.nanmean(L)), "closest/sample", float(nc.mean()))
        S.close()


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "standins":
    main_standins()


# ---- the other integrators behind the seam: small frames rendered by the REFERENCE'S OWN LightTracing::integrate,
# NaivePT::integrate and sub_render_bdpt with one sequential Philox stream (oracle/ref_harness.cpp: tor_render_integrator)
def integrator_cases():
    return {
        "cornell": (lambda: scenes.cornell_box(40, 30), 41),
        "cornell_ggxR_glass": (lambda: scenes.cornell_box(
            40, 30, tall=make_material(MICROFACET_R, (0.725, 0.71, 0.68), roughness=0.3, metallic=0.5),
            short=make_material(PERFECT_REFRACTIVE, (0.725, 0.71, 0.68), eta=1.5)), 42),
        "cornell_ggxT_mirror": (lambda: scenes.cornell_box(
            40, 30, tall=make_material(MICROFACET_T, (0.725, 0.71, 0.68), eta=1.5, roughness=0.2),
            short=make_material(PERFECT_REFLECTIVE, (0.725, 0.71, 0.68))), 43),
        "veach_slight": (lambda: scenes.veach_room(40, 30, small_light=True), 44),
        "cornell_textured": (lambda: scenes.cornell_textured(40, 30), 45),
    }


INTEGRATOR_TYPES = {"light": 1, "naivept": 2, "bdpt": 3}
INTEGRATOR_SPP = 4


def main_integrators():
    R = Oracle("reference")
    out = {}
    for name, (mk, key1) in integrator_cases().items():
        S = R.scene(mk())
        for iname, itype in INTEGRATOR_TYPES.items():
            img = S.render_integrator(itype, INTEGRATOR_SPP, pc.KEY0, key1)
            out[f"{name}.{iname}"] = img
            print(name, iname, "mean", float(img.mean()), "nan", int(np.isnan(img).sum()))
        S.close()
    np.savez_compressed(os.path.join(GOLD, "integrators.npz"), **out)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "integrators":
    main_integrators()
