"""CPU: the port (oracle/tutu_oracle.cpp) against the golden vectors that the REFERENCE build produced
(oracle/gen_golden.py).  Bit-exact: both are plain IEEE fp32 with -ffp-contract=off and the same libm."""
import numpy as np
import pytest

from conftest import golden_path
from helpers import assert_dict_bit_equal, bit_equal
from oracle import parity_cases as pc


def _sub(z, prefix):
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}


@pytest.fixture(scope="module")
def fn_golden():
    return np.load(golden_path("functions.npz"))


def test_philox_known_answers(port, fn_golden):
    z = fn_golden
    for i in range(3):
        out = port.philox(z["philox.ctr"][i:i + 1], int(z["philox.key"][i][0]), int(z["philox.key"][i][1]))[0]
        assert bit_equal(out, z["philox.out"][i])
    # Random123 kat_vectors, philox4x32_10
    assert [hex(x) for x in z["philox.out"][0]] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert [hex(x) for x in z["philox.out"][1]] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    assert [hex(x) for x in z["philox.out"][2]] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]
    xi = port.rng_stream(12345, 7, pc.KEY0, 2, 64)
    assert bit_equal(xi, z["philox.stream"])
    assert (xi >= 0).all() and (xi < 1).all()


def test_bbox(port, fn_golden):
    got = pc.run_bbox(port)
    want = _sub(fn_golden, "bbox.")
    assert_dict_bit_equal(got, want, "bbox.")
    assert 0.05 < want["hit"].mean() < 0.95


def test_triangle(port, fn_golden):
    got = pc.run_tri(port)
    want = _sub(fn_golden, "tri.")
    assert_dict_bit_equal(got, want, "tri.")
    assert 0.2 < want["hit"].mean() < 0.9


def test_math(port, fn_golden):
    assert_dict_bit_equal(pc.run_math(port), _sub(fn_golden, "math."), "math.")


def test_texture_lookup(port, fn_golden):
    """Texture::getRGBat (Texture.hpp:18-39): wrap of negative / >1 coordinates, truncation, index clamp"""
    from tuturenderer_amd import scenes

    assert_dict_bit_equal(pc.run_texture(port, scenes.procedural_maps()["diffuse"][0]), _sub(fn_golden, "texture."), "texture.")


def test_postprocess(port, fn_golden):
    """Postprocessor.hpp under HDR_BLOOM: emissive extraction, separable Gaussian, add, exposure tone map -- including
    the u = 0 -> 1 quirk of the getRGBat reads in column 0 / row 0"""
    assert_dict_bit_equal(pc.run_postprocess(port), _sub(fn_golden, "post."), "post.")


@pytest.mark.parametrize("name", [n for n, _ in pc.material_set()])
def test_material(port, fn_golden, name):
    mat = dict(pc.material_set())[name]
    got = pc.run_material(port, name, mat)
    want = _sub(fn_golden, name + ".")
    assert_dict_bit_equal({k[len(name) + 1:]: v for k, v in got.items()}, want, name + ".")


def test_cornell_restated_matches_reference_obj_load():
    """tuturenderer_amd.scenes.cornell_box() == what objl::Loader + PPMGenerator::loadObj give for
    model/cornellBox/*.obj (vertex order and the generated, un-normalised face normals)."""
    from tuturenderer_amd import scenes

    z = np.load(golden_path("obj_cornell.npz"))
    sc = scenes.cornell_box(64, 64)
    off = 0
    for name, v in scenes.cornell_parts():
        assert bit_equal(v, z[f"cornell_{name}_verts"]), name
        assert bit_equal(sc["normals"][off:off + len(v)], z[f"cornell_{name}_normals"]), name
        off += len(v)
    assert off == 32


SCENES = ["cornell", "cornell_ggxT_mirror", "cornell_ggxR_glass", "veach", "veach_slight", "cornell_degenerate", "cornell_textured", "cornell_spheres"]


def _check_samples(got, want, name):
    """Radiance bit-exact.  The draw / ray counters are bit-exact too except after a refractive vertex with
    pdf < MIN_DIVISOR: the reference traces that continuation and then discards it (PathTracing.hpp:128-133), the
    port does not trace what is discarded -- same radiance, fewer draws."""
    assert bit_equal(got["L"], want["L"])
    assert bit_equal(got["in_crc"], want["in_crc"])
    for k in ("ndraws", "nclosest"):
        diff = got[k] != want[k]
        if name == "cornell":
            assert not diff.any()
        else:
            assert (got[k] <= want[k]).all() and diff.mean() < 0.02


@pytest.mark.parametrize("name", SCENES)
def test_scene_and_samples(port, name):
    from oracle.gen_golden import golden_scenes

    mk, key1 = golden_scenes()[name]
    z = np.load(golden_path(f"scene_{name}.npz"))
    S = port.scene(mk())
    b, leaf = S.bvh_dump()
    assert bit_equal(b, z["bvh.bounds"]) and bit_equal(leaf, z["bvh.leaf_tri"])
    assert_dict_bit_equal(pc.run_scene(S), _sub(z, "scene."), "scene.")
    _check_samples(pc.run_samples(S, key1), _sub(z, "samples."), name)
    # ordered, t-pruned traversal of the same tree (the HIP kernels' algorithm) gives the same answers
    S.lib.tor_port_set_ordered(S.h, 1)
    assert_dict_bit_equal(pc.run_scene(S), _sub(z, "scene."), "scene(ordered).")
    _check_samples(pc.run_samples(S, key1), _sub(z, "samples."), name)
    S.close()


def test_render_image(port):
    from oracle.gen_golden import golden_scenes

    mk, key1 = golden_scenes()["cornell"]
    z = np.load(golden_path("scene_cornell.npz"))
    S = port.scene(mk())
    img = S.render(16, pc.KEY0, key1, nthreads=4)
    assert bit_equal(img, z["render.rgb"])
    # a sub-rectangle renders the same pixels (tiles are independent: counter-based RNG)
    part = S.render(16, pc.KEY0, key1, rect=(10, 20, 50, 40), nthreads=2)
    assert bit_equal(part[20:40, 10:50], z["render.rgb"][20:40, 10:50])
    S.close()


@pytest.mark.parametrize("name", ["bunny", "broom"])
def test_port_matches_reference_on_standin_scenes(port, name):
    """BASELINE configs 3 / 4 stand-ins: the CPU restatement against what the reference build answered (4000 samples'
    radiance from its own traceRay, closest hits of the primary rays): bit for bit, like the other scenes."""
    from oracle.gen_golden import standin_sample_ids, standin_scenes

    mk, key1 = standin_scenes()[name]
    sc = mk()
    z = np.load(golden_path(f"scene_{name}.npz"))
    assert pc.checksum(np.ascontiguousarray(sc["verts"], np.float32), np.ascontiguousarray(sc["normals"], np.float32)) == z["scene.crc"]
    pix, smp = standin_sample_ids(sc)
    S = port.scene(sc)
    L, nd, nc = S.trace_samples(pix, smp, pc.KEY0, key1, stats=True)
    d = S.raydir((pix % sc["width"]).astype(np.int32), (pix // sc["width"]).astype(np.int32))
    o = np.repeat(S.camera()[5][None], len(pix), 0)
    hit, t, tri, *_ = S.closest(o, d)
    S.close()
    assert bit_equal(tri, z["primary.tri"]) and bit_equal(np.where(hit.astype(bool), t, 0).astype(np.float32), z["primary.t"])
    assert bit_equal(L, z["samples.L"])
    # work counters: the reference traces the continuation of a refractive vertex BEFORE it tests pdf < MIN_DIVISOR and then
    # drops the result (PathTracing.hpp:128-133); the restatement stops there -- fewer draws / rays on those samples, same radiance
    assert (nd <= z["samples.ndraws"]).all() and (nc <= z["samples.nclosest"]).all()
    assert (nd != z["samples.ndraws"]).mean() < 0.01


@pytest.mark.parametrize("iname", ["light", "naivept", "bdpt"])
def test_other_integrators_match_reference_frames(port, iname):
    """the restatement of LightTracing / NaivePT / BDPT against frames rendered by the reference's OWN
    LightTracing::integrate, NaivePT::integrate and sub_render_bdpt (tests/golden/integrators.npz, `gen_golden.py
    integrators`): same loop order, one sequential Philox stream -- bit for bit, on five scenes incl. rough glass, mirror,
    the veach room and textures"""
    from oracle.gen_golden import INTEGRATOR_SPP, INTEGRATOR_TYPES, integrator_cases

    z = np.load(golden_path("integrators.npz"))
    for name, (mk, key1) in integrator_cases().items():
        S = port.scene(mk())
        img = S.render_integrator(INTEGRATOR_TYPES[iname], INTEGRATOR_SPP, pc.KEY0, key1)
        S.close()
        assert bit_equal(img, z[f"{name}.{iname}"]), (name, iname, float(np.abs(img - z[f"{name}.{iname}"]).max()))
        assert np.isfinite(img).all()


# ---- the whole-frame fixtures of the BASELINE configs (oracle/gen_frames.py: the REFERENCE build's frames) ------------------
FRAME_RECTS = {  # name -> (scene maker, a 24 x 4 pixel rectangle that sees geometry)
    "c2": (lambda s: s.cornell_box(800, 800), (380, 500)),
    "c5": (lambda s: s.veach_room(800, 600, small_light=False), (300, 330)),
    "c3": (lambda s: s.bunny_box(1024, 1024), (560, 700)),
    "c4": (lambda s: s.broom_room(1600, 900), (800, 560)),
    "c1": (lambda s: s.cornell_box(800, 800), (380, 500)),          # round 5: BASELINE configs[0], 16 spp
    "c4_64": (lambda s: s.broom_room(1600, 900), (800, 560)),       # round 5: the broom frame at 64 spp
}


@pytest.mark.parametrize("name", sorted(FRAME_RECTS))
def test_port_reproduces_a_rectangle_of_the_reference_builds_frame(port, name):
    """tests/golden/frame_<config>.npz holds what the reference's own traceRay left in the frame buffer at BASELINE size and
    spp (c4: 16 spp).  The CPU restatement renders 96 of those pixels at the fixture's spp and key: the same bits."""
    from tuturenderer_amd import scenes

    mk, (x0, y0) = FRAME_RECTS[name]
    z = np.load(golden_path(f"frame_{name}.npz"))
    sc = mk(scenes)
    assert z["rgb"].shape == (sc["height"], sc["width"], 3) and z["rgb"].dtype == np.float32
    assert pc.checksum(np.ascontiguousarray(sc["verts"], np.float32), np.ascontiguousarray(sc["normals"], np.float32),
                       np.ascontiguousarray(sc["mat_id"], np.int32)) == z["scene_crc"]
    S = port.scene(sc)
    got = S.render(int(z["spp"]), int(z["key0"]), int(z["key1"]), rect=(x0, y0, x0 + 24, y0 + 4), nthreads=8)
    S.close()
    want = z["rgb"][y0:y0 + 4, x0:x0 + 24]
    assert want.mean() > 1e-3  # the rectangle is not black
    assert bit_equal(got[y0:y0 + 4, x0:x0 + 24], want)
