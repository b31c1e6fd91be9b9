"""GPU (MI355X): the HIP path, called through the C ABI, against the CPU oracle and the golden vectors that the
reference build produced.

Bars (round 5: with the C library's sinf / cosf / acosf / tanf / powf restated for the device, csrc/device_libm.h):
  * traversal decisions (hit / triangle index / blocked), every +,-,*,/,sqrt-only quantity (t, barycentrics, light samples,
    camera rays) AND every material function (BxDF, pdf, sampleDirection, Fresnel, D, G): BIT-EXACT;
  * per-sample radiance at matched Philox seed: every sample within 5e-6 relative of the reference build's (measured: 4e-7 =
    a few ulp; 13-34 % of the samples differ in a last bit).  What is left is the order of one sum: the reference's traceRay
    is a recursion that adds from the tail, L_d = S_d + coe_d * L_{d+1} (PathTracing.hpp:275-277), the wavefront carries the
    product of the coe's forward and adds S_d * beta_d as it goes -- the same real number, rounded in another order;
  * images at matched seed: mean per-pixel L2 < 1e-6 (measured 3e-9 .. 1e-8; the tolerance BASELINE.json's north_star states
    is 1e-3).
"""
import numpy as np
import pytest

from conftest import golden_path
from helpers import bit_equal, count_diff, ulp_diff
from oracle import parity_cases as pc

pytestmark = pytest.mark.gpu

SCENES = ["cornell", "cornell_ggxT_mirror", "cornell_ggxR_glass", "veach", "veach_slight", "cornell_degenerate", "cornell_textured", "cornell_spheres"]


@pytest.fixture(scope="module")
def tr(built):
    import tuturenderer_amd

    tuturenderer_amd.load_library()
    assert tuturenderer_amd.device_count() >= 1, "no HIP device: the product path has no fallback"
    return tuturenderer_amd


def _scene(name):
    from oracle.gen_golden import golden_scenes

    mk, key1 = golden_scenes()[name]
    return mk(), key1


class HipSceneAdapter:
    """gives a tuturenderer_amd.Context the method names parity_cases.run_scene expects"""

    def __init__(self, ctx, oracle_scene):
        self.ctx = ctx
        self.S = oracle_scene
        self.W, self.H, self.verts = oracle_scene.W, oracle_scene.H, oracle_scene.verts

    def raydir(self, px, py):
        return self.S.raydir(px, py)

    def camera(self):
        return self.S.camera()


@pytest.mark.parametrize("name", SCENES)
def test_closest_and_any_hit_bit_exact(tr, port, name):
    sc, _ = _scene(name)
    z = np.load(golden_path(f"scene_{name}.npz"))
    S = port.scene(sc)
    O, D = pc.scene_rays(S)
    with tr.Context(sc) as ctx:
        hits = ctx.trace_closest(O, D)
        h = hits["tri"] >= 0
        assert bit_equal(h.astype(np.uint8), z["scene.hit"])
        assert bit_equal(hits["tri"], z["scene.tri"])
        assert bit_equal(np.where(h, hits["t"], 0).astype(np.float32), z["scene.t"])
        # shadow queries exactly as parity_cases.run_scene builds them
        want = pc.run_scene(S)
        r = pc._rng(105 + 1)
        idx = np.nonzero(h)[0]
        xi = pc.xi24(r, (len(idx), 3))
        ltri, lpos, lnrm, lpdf = ctx.eval_sample_light(xi)
        assert bit_equal(ltri, z["scene.light_tri"]) and bit_equal(lpdf, z["scene.light_pdf"])
        # a point on a sphere light goes through cosf/sinf (Sphere.hpp:152-158): a few ulp; triangle lights are exact
        on_sphere = np.isin(ltri, np.asarray(sc.get("sphere_pos", []), np.int32))
        assert bit_equal(lpos[~on_sphere], z["scene.light_pos"][~on_sphere]) and bit_equal(lnrm[~on_sphere], z["scene.light_nrm"][~on_sphere])
        if on_sphere.any():
            assert np.abs(lpos[on_sphere] - z["scene.light_pos"][on_sphere]).max() < 1e-4
            assert np.abs(lnrm[on_sphere] - z["scene.light_nrm"][on_sphere]).max() < 1e-5
        lpos, lnrm = z["scene.light_pos"], z["scene.light_nrm"]  # shadow targets: the reference's own points
        pos, Ns = want["pos"], want["Ns"]
        orig = (pos[idx] + np.float32(0.0005) * Ns[idx] * np.sign((Ns[idx] * -D[idx]).sum(1, keepdims=True))).astype(np.float32)
        target = (lpos + np.float32(0.0005) * lnrm).astype(np.float32)
        half = len(idx) // 2
        target[half:] = pos[idx][::-1][half:]
        blocked = ctx.trace_any(orig, target)
        assert bit_equal(blocked, z["scene.blocked"])
        assert 0.02 < blocked.mean() < 0.98
    S.close()


def test_closest_hit_many_random_rays_vs_oracle(tr, port):
    """1e6 rays on the veach room (4615 nodes, depth 12): identical triangle and identical t bits."""
    sc, _ = _scene("veach_slight")
    S = port.scene(sc)
    r = pc._rng(909)
    n = 1_000_000
    lo = S.verts.reshape(-1, 3).min(0)
    hi = S.verts.reshape(-1, 3).max(0)
    o = (lo + (hi - lo) * r.random((n, 3))).astype(np.float32)
    d = pc.unit(r, n)
    with tr.Context(sc) as ctx:
        hits = ctx.trace_closest(o, d)
    hit, t, tri, *_ = S.closest(o, d)
    assert count_diff(hits["tri"], tri) == 0
    assert count_diff(np.where(tri >= 0, hits["t"], 0), np.where(tri >= 0, t, 0)) == 0
    S.close()


@pytest.mark.parametrize("name", ["cornell", "veach_slight", "cornell_spheres"])
def test_degenerate_rays_take_the_reference_tree(tr, port, name):
    """Rays with a zero direction component (1/d = inf: the slab products can be NaN, BoundBox.hpp:57-62) or starting
    exactly on box planes do not walk the SAH tree but the reference's own tree: same hit, same object, same t bits,
    same shadow answers as the reference's unordered recursion."""
    sc, _ = _scene(name)
    S = port.scene(sc)
    r = pc._rng(4242)
    n = 60000
    V = S.verts.reshape(-1, 3)
    lo, hi = V.min(0), V.max(0)
    # origins: random, but a third of the coordinates snapped onto vertex coordinates (= box planes of the tree)
    o = (lo + (hi - lo) * r.random((n, 3))).astype(np.float32)
    snap = r.random((n, 3)) < 0.33
    o = np.where(snap, V[r.integers(0, len(V), (n, 3)), np.arange(3)], o).astype(np.float32)
    d = pc.unit(r, n)
    kind = r.integers(0, 4, n)
    axis = r.integers(0, 3, n)
    for k in range(3):
        d[(kind == 0) & (axis == k), k] = 0.0                    # one zero component
        d[(kind == 1) & (axis != k), k] = 0.0                    # axis-parallel: two zero components
    d[kind == 2, 0] = np.float32(-0.0)                           # negative zero
    nz = np.linalg.norm(d, axis=1) > 0
    o, d = o[nz], d[nz]
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    hit, t, tri, *_ = S.closest(o, d)
    target = (o + d * np.float32(200.0) * r.random((len(o), 1)).astype(np.float32)).astype(np.float32)
    blocked = S.any_hit(o, target)
    with tr.Context(sc) as ctx:
        hits = ctx.trace_closest(o, d)
        got_blocked = ctx.trace_any(o, target)
    assert count_diff(hits["tri"], tri) == 0
    assert count_diff(np.where(tri >= 0, hits["t"], 0), np.where(tri >= 0, t, 0)) == 0
    assert count_diff(got_blocked, blocked) == 0
    assert 0.2 < (tri >= 0).mean() and 0.02 < blocked.mean() < 0.98
    S.close()


def test_postprocess_and_quantise(tr, port):
    """Postprocessor.hpp (HDR_BLOOM) and writePixel's gamma mapping on the device against the reference build's
    outputs: emissive extraction, Gaussian blur and the add are + - * / sqrt only -> bit-exact; the tone map goes
    through exp (formed in double, rounded once) and the quantisation through pow -> equal except for isolated
    last-bit / level differences."""
    from tuturenderer_amd import scenes

    z = np.load(golden_path("functions.npz"))
    img = pc.postprocess_input()
    with tr.Context(scenes.cornell_box(8, 8)) as ctx:
        assert bit_equal(ctx.postprocess(img, 1), z["post.emissive"])
        assert bit_equal(ctx.postprocess(img, 2), z["post.blur"])
        for stage, key in ((3, "post.hdr"), (0, "post.full")):
            got = ctx.postprocess(img, stage)
            # 1 - exp(..): one ulp of an exp() near 1 is 6e-8 absolute, many ulps of the small difference
            assert (got != z[key]).mean() < 2e-3 and np.abs(got - z[key]).max() <= 1.2e-7, key
        # a full frame: the reference's post-process of a rendered image
        frame = ctx.render(4, pc.KEY0, 1)
        c = np.linspace(-0.2, 1.3, 100001, dtype=np.float32)
        lv = ctx.quantise(c)
    want = port.write_pixel(c)
    assert (lv != want).mean() < 1e-4 and np.abs(lv - want).max() <= 1
    assert lv.min() == 0 and lv.max() == 255
    with tr.Context(scenes.cornell_box(64, 64)) as ctx:
        frame = ctx.render(4, pc.KEY0, 1)
        got = ctx.postprocess(frame, 0)
    ref = port.postprocess(0, frame)
    assert (got != ref).mean() < 2e-3 and np.abs(got - ref).max() <= 1.2e-7


@pytest.mark.parametrize("name", [n for n, _ in pc.material_set()])
def test_material_functions(tr, name):
    mat = dict(pc.material_set())[name]
    zf = np.load(golden_path("functions.npz"))
    want = {k[len(name) + 1:]: zf[k] for k in zf.files if k.startswith(name + ".")}

    class Dev:
        def __init__(self, ctx, port):
            self.ctx, self.port = ctx, port

        # deterministic helpers used to build inputs come from the port (they are inputs, not the thing tested)
        def normalized(self, v):
            return self.port.normalized(v)

        def reflect(self, a, b):
            return self.port.reflect(a, b)

        def refract(self, a, b, c, d):
            return self.port.refract(a, b, c, d)

        def mat_bxdf(self, *a, **k):
            return self.ctx.eval_bxdf(*a, **k)

        def mat_pdf(self, *a, **k):
            return self.ctx.eval_pdf(*a, **k)

        def mat_sample(self, *a, **k):
            return self.ctx.eval_sample(*a, **k)

    from oracle.pyoracle import Oracle
    from tuturenderer_amd import scenes

    with tr.Context(scenes.cornell_box(8, 8)) as ctx:
        got = pc.run_material(Dev(ctx, Oracle("port")), name, mat, s_wi_at=want["s_wi"])
    got = {k[len(name) + 1:]: v for k, v in got.items()}
    # Round 5: every output is the reference build's, bit for bit (until then the values that go through sinf / cosf / acosf / tanf /
    # x^5 were held to 16 ulp, and a discrete outcome was allowed to flip where xi sat within an ulp of F).
    for k in ("s_ok", "s_special", "s_ndraws"):
        assert count_diff(got[k], want[k]) == 0, k
    for k in ("bxdf", "bxdf_tir", "pdf", "s_wi", "s_pdf", "s_bxdf"):
        g, w = np.asarray(got[k], np.float32), np.asarray(want[k], np.float32)
        bad = (g.view(np.uint32) != w.view(np.uint32)) & ~(np.isnan(g) & np.isnan(w))
        assert not bad.any(), (k, int(bad.sum()), g[bad][:4], w[bad][:4])


def test_texture_lookup_bit_exact(tr):
    """Texture::getRGBat on the device against the reference build's answers (wrap, truncation, index clamp)"""
    from tuturenderer_amd import scenes

    z = np.load(golden_path("functions.npz"))
    sc = scenes.cornell_textured(32, 32)
    u, v = pc.texture_inputs()
    with tr.Context(sc) as ctx:
        got = ctx.eval_texture(0, 0, u, v)
        assert bit_equal(got, z["texture.rgb"])
        # every map of every list answers, and an index beyond a list is refused like the reference's exit(1)
        for k, name in enumerate(("diffuse", "normal", "roughness", "metallic")):
            for i, img in enumerate(sc["textures"][name]):
                g = ctx.eval_texture(k, i, u[:64], v[:64])
                assert np.isin(g.reshape(-1), np.asarray(img, np.float32).reshape(-1)).all()
            with pytest.raises(tr.TutuError):
                ctx.eval_texture(k, len(sc["textures"][name]), u[:4], v[:4])
    bad = dict(sc)
    bad["tex_ids"] = sc["tex_ids"].copy()
    bad["tex_ids"][0, 0] = 7
    with pytest.raises(tr.TutuError):
        tr.Context(bad)


def test_textured_edge_cases(tr, port):
    """degenerate uvs under a normal map (coef = 1/0 -> NaN tangent frame -> NaN shading normal: every comparison on
    it fails and the reference returns black there), default (-1,-1) uvs under an albedo map, an empty 0x0 map"""
    from tuturenderer_amd import scenes

    sc = scenes.cornell_textured(48, 48)
    sc["uvs"] = sc["uvs"].copy()
    sc["tex_ids"] = sc["tex_ids"].copy()
    floor = np.nonzero(sc["tex_ids"][:, 1] == 0)[0]
    sc["uvs"][floor[0]] = 0.25               # all three corners equal: degenerate tangent frame
    sc["uvs"][floor[1]] = -1.0               # Vector2f() default
    sc["textures"] = dict(sc["textures"])
    sc["textures"]["diffuse"] = list(sc["textures"]["diffuse"]) + [np.zeros((0, 0, 3), np.float32)]
    short = np.nonzero(sc["tex_ids"][:, 0] == 1)[0]
    sc["tex_ids"][short[:4], 0] = 2          # the empty map: reads as black
    S = port.scene(sc)
    pix, smp = pc.sample_ids(S, n=6000)
    want = S.trace_samples(pix, smp, pc.KEY0, 21)
    with tr.Context(sc) as ctx:
        L = ctx.trace_samples(pix, smp, pc.KEY0, 21)
        img = ctx.render(8, pc.KEY0, 21)
    ref = S.render(8, pc.KEY0, 21, nthreads=4)
    S.close()
    nan_g, nan_w = np.isnan(L).any(1), np.isnan(want).any(1)
    assert (nan_g != nan_w).mean() < 1e-3
    fin = ~(nan_g | nan_w)
    err = np.abs(L[fin] - want[fin]).max(1)
    scale = np.maximum(np.abs(want[fin]).max(1), 1e-3)
    assert (err > 1e-4 * scale + 1e-6).mean() < 5e-3
    assert (want == 0).all(1).mean() > 0.05  # the black triangle is in view
    assert np.isfinite(img).all()
    assert np.sqrt(((img - ref) ** 2).sum(-1)).mean() < 1e-3


@pytest.mark.parametrize("name", SCENES)
def test_per_sample_radiance_matched_seed(tr, port, name):
    sc, key1 = _scene(name)
    z = np.load(golden_path(f"scene_{name}.npz"))
    S = port.scene(sc)
    pix, smp = pc.sample_ids(S)
    with tr.Context(sc) as ctx:
        L = ctx.trace_samples(pix, smp, pc.KEY0, key1)
    want = z["samples.L"]  # produced by the reference's own traceRay on the same Philox stream
    # NaN samples (the zero-area light of the degenerate scene makes ~1 in 6) must be NaN on both sides
    nan_g, nan_w = np.isnan(L).any(1), np.isnan(want).any(1)
    assert (nan_g != nan_w).mean() < 1e-3, (int(nan_g.sum()), int(nan_w.sum()))
    if name == "cornell_degenerate":
        assert nan_w.mean() > 0.05
    fin = ~(nan_g | nan_w)
    err = np.abs(L[fin] - want[fin]).max(1)
    scale = np.maximum(np.abs(want[fin]).max(1), 1e-3)
    ok = err <= 5e-6 * scale + 1e-7
    bits = (L[fin].view(np.uint32) != want[fin].view(np.uint32)).any(1)
    print(f"{name}: NaN samples {int(nan_w.sum())}; beyond 5e-6 relative {int((~ok).sum())}/{len(ok)}; max rel err {float((err / scale).max()):.2e}; "
          f"samples with a differing last bit {bits.mean():.3f}; mean |dL| {float(err.mean()):.3e}")
    # Until round 5, 0.1-1 % of the samples took another path than the reference's (a sampled direction one ulp off, from the
    # device's own sinf / cosf, moved a hit across an edge or flipped a shadow ray at a sphere light's own surface).  None does
    # now: the differences left are the rounding of the radiance sum (module docstring).
    assert ok.all(), (int((~ok).sum()), float((err / scale).max()))
    assert (nan_g == nan_w).all()
    assert float(err.mean()) < 1e-6
    assert abs(L[fin].mean() - want[fin].mean()) < 2e-2 * max(want[fin].mean(), 1e-3)
    S.close()


@pytest.mark.parametrize("name", SCENES)
def test_per_sample_radiance_exact_sum_is_the_reference_builds(tr, port, name):
    """knob exact_sum (device_shade.h: PassParams::xlog): the radiance of every sample of every golden scene -- Lambertian, mirror, glass,
    both microfacet models, textures, spheres, the degenerate light whose samples are NaN -- is the reference build's traceRay's, bit
    for bit (NaN where it is NaN)."""
    sc, key1 = _scene(name)
    z = np.load(golden_path(f"scene_{name}.npz"))
    S = port.scene(sc)
    pix, smp = pc.sample_ids(S)
    S.close()
    with tr.Context(sc) as ctx:
        ctx.set_option("exact_sum", 1)
        L = ctx.trace_samples(pix, smp, pc.KEY0, key1)
        assert ctx.get_option("exact_sum") == 1
    want = z["samples.L"]
    bad = ((L.view(np.uint32) != want.view(np.uint32)) & ~(np.isnan(L) & np.isnan(want))).any(1)
    assert not bad.any(), (int(bad.sum()), L[bad][:3], want[bad][:3])


@pytest.mark.parametrize("name", ["cornell", "cornell_ggxR_glass", "veach_slight", "cornell_degenerate", "cornell_textured", "cornell_spheres"])
def test_image_matched_seed_l2(tr, name):
    sc, key1 = _scene(name)
    z = np.load(golden_path(f"scene_{name}.npz"))
    ref = z["render.rgb"]  # reference build, 16 spp, same Philox key
    with tr.Context(sc) as ctx:
        img = ctx.render(16, pc.KEY0, key1)
        st = ctx.last_stats
    l2 = np.sqrt(((img - ref) ** 2).sum(-1))
    print(f"{name}: mean per-pixel L2 {l2.mean():.3e}  pixels with L2>1e-2: {(l2 > 1e-2).mean():.2e}  stats {st}")
    assert np.isfinite(img).all()
    assert l2.mean() < 1e-6  # (north_star: 1e-3)
    assert st["samples"] == img.shape[0] * img.shape[1] * 16


def test_render_is_independent_of_pass_size_tiling_and_worklist(tr):
    """Counter-based RNG keyed by (pixel, sample): the image must not depend on how the work is batched.
    Same arithmetic, same order per pixel -> bit-identical."""
    sc, key1 = _scene("cornell_ggxT_mirror")
    with tr.Context(sc) as ctx:
        a = ctx.render(12, pc.KEY0, key1)
        b = ctx.render(12, pc.KEY0, key1, spp_per_pass=5)  # ragged last pass
        c = ctx.render(12, pc.KEY0, key1, max_paths=1000)   # many tiny passes
        assert bit_equal(a, b) and bit_equal(a, c)
        # a sub-rectangle and an explicit (shuffled, interleaved) pixel list give the same pixels
        part = ctx.render(12, pc.KEY0, key1, rect=(10, 20, 50, 40))
        assert bit_equal(part[20:40, 10:50], a[20:40, 10:50])
        assert not part[:20].any() and not part[:, :10].any()
        rng = np.random.default_rng(3)
        pixels = rng.permutation(ctx.W * ctx.H)[:777].astype(np.int32)
        lst = ctx.render(12, pc.KEY0, key1, pixels=pixels, full_frame=False)
        assert bit_equal(lst, a.reshape(-1, 3)[pixels])
        # a different key changes the image
        d = ctx.render(12, pc.KEY0, key1 + 1)
        assert not bit_equal(a, d)


def test_image_does_not_depend_on_passes_in_flight(tr):
    """tutu_hip_set_option("sets"): one to four wavefront passes in flight on as many streams -- the resolves are
    ordered by events, so the frame is bit-identical"""
    sc, key1 = _scene("cornell_ggxR_glass")
    with tr.Context(sc) as ctx:
        ref = ctx.render(24, pc.KEY0, key1, max_paths=4 * 96 * 96 * 3)  # 3 spp per pass, 8 passes
        for sets in (1, 2, 3, 4):
            ctx.set_option("sets", sets)
            img = ctx.render(24, pc.KEY0, key1, max_paths=4 * 96 * 96 * 3)
            assert bit_equal(img, ref), sets
            assert ctx.last_stats["passes"] == 8
        with pytest.raises(tr.TutuError):
            ctx.set_option("sets", 9)
        with pytest.raises(tr.TutuError):
            ctx.set_option("no_such_option", 1)


def test_edge_cases(tr):
    from tuturenderer_amd import scenes

    # empty scene: every pixel is the background colour
    sc = scenes.cornell_box(16, 16)
    empty = dict(sc)
    empty["verts"] = np.zeros((0, 9), np.float32)
    empty["normals"] = np.zeros((0, 9), np.float32)
    empty["mat_id"] = np.zeros((0,), np.int32)
    empty["bkg"] = (0.25, 0.5, 0.75)
    with tr.Context(empty) as ctx:
        img = ctx.render(3, 1, 2)
        assert np.allclose(img, np.array([0.25, 0.5, 0.75], np.float32))
    # a scene without lights: black (no NEE, BSDF rays find no emitter)
    nolight = dict(sc)
    nolight["mats"] = sc["mats"].copy()
    nolight["mats"]["emission"] = 0
    with tr.Context(nolight) as ctx:
        assert ctx.info()["n_lights"] == 0
        assert not ctx.render(4, 1, 2).any()
    # single triangle, single pixel, 1 spp; bad arguments are rejected with codes
    one = dict(sc)
    one["verts"], one["normals"], one["mat_id"] = sc["verts"][:1], sc["normals"][:1], sc["mat_id"][:1]
    one["width"] = one["height"] = 1
    with tr.Context(one) as ctx:
        assert ctx.render(1, 1, 2).shape == (1, 1, 3)
        with pytest.raises(tr.TutuError):
            ctx.render(0, 1, 2)
        with pytest.raises(tr.TutuError):
            ctx.render(1, 1, 2, rect=(0, 0, 2, 1))
        with pytest.raises(tr.TutuError):
            ctx.render(1, 1, 2, pixels=[5])


def test_full_size_properties(tr):
    """BASELINE config sizes are too big for the CPU oracle in a test; check size-independent properties:
    (a) linearity in emission (radiance scales exactly by a power of two), (b) an all-black-albedo box shows only
    directly visible emission, (c) 800x800 at 64 spp equals the mean of its two 32-spp halves' sample sets."""
    from tuturenderer_amd import scenes

    sc = scenes.cornell_box(800, 800)
    with tr.Context(sc) as ctx:
        a = ctx.render(8, 7, 8)
        st = ctx.last_stats
        assert st["samples"] == 800 * 800 * 8 and st["closest_rays"] > st["samples"]
    sc2 = scenes.cornell_box(800, 800)
    sc2["mats"] = sc2["mats"].copy()
    sc2["mats"]["emission"] *= np.float32(4.0)
    with tr.Context(sc2) as ctx:
        b = ctx.render(8, 7, 8)
    assert bit_equal(b, a * np.float32(4.0))  # every path is linear in Le; x4 is exact in fp32
    sc3 = scenes.cornell_box(800, 800)
    sc3["mats"] = sc3["mats"].copy()
    sc3["mats"]["diffuse"] = 0
    with tr.Context(sc3) as ctx:
        c = ctx.render(2, 7, 8)
    lit = c.sum(-1) > 0
    assert 0.001 < lit.mean() < 0.05  # only the light's own pixels
    assert np.allclose(c[lit], np.array(scenes.CB_EMISSION, np.float32))


def test_sliver_mesh_reference_cap_and_threaded_build(tr, monkeypatch):
    """90 000 long thin triangles at random angles: early split clipping gives each ~14 references; the scene's extra references
    are capped (4 Mi by default, 1 Mi here: the per-triangle limit is halved until they fit).  The walked tree is built on
    several host threads; it is the tree one thread builds, and the hits are the reference tree's, bit for bit."""
    from tuturenderer_amd import scenes

    rng = np.random.default_rng(9)
    n = 90000
    c = rng.uniform(0, 100, (n, 1, 3)).astype(np.float32)
    d = rng.normal(size=(n, 1, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    w = rng.normal(size=(n, 1, 3)).astype(np.float32) * np.float32(0.15)
    verts = np.concatenate([c - 20 * d, c + 20 * d + w, c + 20 * d - w], 1).astype(np.float32)  # (n, 3, 3)
    sc = scenes.cornell_box(64, 64)
    sc = dict(sc, verts=verts.reshape(-1, 9), normals=scenes.face_normals(verts.reshape(-1, 9)), mat_id=np.zeros(n, np.int32))
    o = rng.uniform(0, 100, (200000, 3)).astype(np.float32)
    dd = rng.normal(size=(200000, 3)).astype(np.float32)
    dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    hits = {}
    for tag, env in (("threads", {}), ("serial", {"TUTU_BUILD_SERIAL": "1"}), ("reference_tree", {"TUTU_NO_SAH": "1"})):
        for k in ("TUTU_BUILD_SERIAL", "TUTU_NO_SAH"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        monkeypatch.setenv("TUTU_SPLIT_CAP_MI", "1")
        with tr.Context(sc) as ctx:
            refs = ctx.get_option("n_refs")
            hits[tag] = ctx.trace_closest(o, dd)
        if tag != "reference_tree":
            assert 5 * n < refs <= n + (1 << 20), refs
    for k in ("TUTU_BUILD_SERIAL", "TUTU_NO_SAH", "TUTU_SPLIT_CAP_MI"):
        monkeypatch.delenv(k, raising=False)
    with tr.Context(sc) as ctx:
        assert ctx.get_option("n_refs") > n + (1 << 20)  # without the lowered cap: ~14 references per sliver
    assert (hits["threads"]["tri"] >= 0).mean() > 0.2
    for tag in ("serial", "reference_tree"):
        assert bit_equal(hits["threads"]["tri"], hits[tag]["tri"]) and bit_equal(hits["threads"]["t"], hits[tag]["t"]), tag


@pytest.mark.parametrize("name", ["bunny", "broom"])
def test_stand_in_scenes_per_sample_parity(tr, name):
    """BASELINE configs 3 and 4 (synthetic stand-ins, 82 k / 48 k triangles, BVH depth 17 / 16, rough glass /
    rough metal): these go through the HBM-resident traversal path.  Matched-seed radiance of 4000 random samples and
    the closest hits of their primary rays against what the REFERENCE BUILD answered (tests/golden/scene_bunny.npz,
    scene_broom.npz, written by `oracle/gen_golden.py standins`), at the bar of the other scenes."""
    from oracle.gen_golden import standin_sample_ids, standin_scenes

    mk, key1 = standin_scenes()[name]
    sc = mk()
    z = np.load(golden_path(f"scene_{name}.npz"))
    assert pc.checksum(np.ascontiguousarray(sc["verts"], np.float32), np.ascontiguousarray(sc["normals"], np.float32)) == z["scene.crc"], \
        "the procedural scene is not the one the golden vectors were made for"
    pix, smp = standin_sample_ids(sc)
    assert pc.checksum(pix, smp) == z["samples.in_crc"]
    with tr.Context(sc) as ctx:
        info = ctx.info()
        L = ctx.trace_samples(pix, smp, pc.KEY0, key1)
        ctx.set_option("exact_sum", 1)
        Lx = ctx.trace_samples(pix, smp, pc.KEY0, key1)
        ctx.set_option("exact_sum", 0)
        cam = tr.camera_frame_array(sc)
        x, y = (pix % sc["width"]).astype(np.float32), (pix // sc["width"]).astype(np.float32)
        p = cam[0] + x[:, None] * cam[1] + y[:, None] * cam[2] + cam[4] + cam[4]  # PathTracing.hpp:503 [sic]
        d = p - cam[5]
        d = (d * (np.float32(1) / np.sqrt((d * d).sum(1, dtype=np.float32), dtype=np.float32))[:, None]).astype(np.float32)
        hits = ctx.trace_closest(np.repeat(cam[5][None], len(pix), 0), d)
    # kernel level: same object, same t bits as the reference's getIntersection (the ray directions above may differ from the
    # reference's by an ulp in the normalisation, so a handful of edge rays are allowed to land on a neighbour)
    tri_bad = (hits["tri"] != z["primary.tri"]).mean()
    t_bad = (np.where(hits["tri"] >= 0, hits["t"], 0).astype(np.float32) != z["primary.t"]).mean()
    want = z["samples.L"]
    nan_g, nan_w = np.isnan(L).any(1), np.isnan(want).any(1)
    fin = ~(nan_g | nan_w)
    err = np.abs(L[fin] - want[fin]).max(1)
    scale = np.maximum(np.abs(want[fin]).max(1), 1e-3)
    bad = 1.0 - (err <= 1e-4 * scale + 1e-6).mean()
    print(f"{name}: {info} diverged samples {bad:.4f}, primary tri mismatches {tri_bad:.5f}, t mismatches {t_bad:.5f}")
    assert info["depth"] >= 15
    assert (nan_g != nan_w).mean() < 1e-3
    assert tri_bad < 2e-3 and t_bad < 1e-2
    assert bad == 0.0, bad  # (round 5: no sample takes another path than the reference's)
    assert abs(L[fin].mean() - want[fin].mean()) < 2e-2 * max(want[fin].mean(), 1e-3)
    # ... and folded in the reference's order (knob exact_sum) every sample is the reference build's, bit for bit
    n_bits = int(((Lx.view(np.uint32) != want.view(np.uint32)) & ~(np.isnan(Lx) & np.isnan(want))).any(1).sum())
    assert n_bits == 0, n_bits


# ------------------------------------------------------------------------------------------------------------
# function-level parity of the rows that were green only by implication: a10 slab test, a11 triangle test, a16 math
# helpers, a17 RNG -- the DEVICE functions, through tutu_hip_eval_fn, against the reference build's golden vectors
@pytest.fixture(scope="module")
def fn_ctx(tr):
    from tuturenderer_amd import scenes

    with tr.Context(scenes.cornell_box(8, 8)) as ctx:
        yield ctx


def test_device_slab_test_bit_exact(fn_ctx):
    """BoundBox::IntersectRay on the device: 20 000 boxes / rays incl. flat boxes, zero and negative-zero direction
    components, origins on slab planes (NaN products fall through the ?: selects exactly as in the reference)"""
    z = np.load(golden_path("functions.npz"))
    pmin, pmax, o, d = pc.bbox_inputs()
    assert pc.checksum(pmin, pmax, o, d) == z["bbox.in_crc"]
    got = fn_ctx.eval_fn("bbox", pmin, pmax, o, d)[:, 0]
    assert count_diff(got.astype(np.uint8), z["bbox.hit"]) == 0
    assert 0.05 < got.mean() < 0.9


def test_device_triangle_test(fn_ctx):
    """Triangle::intersect on the device (host-hoisted E1, E2, normal): hit decisions and t bit-exact; pos, Ns, Ng bit-exact
    on the hits (+ - * / sqrt only)"""
    z = np.load(golden_path("functions.npz"))
    verts, normals, o, d = pc.tri_inputs()
    assert pc.checksum(verts, normals, o, d) == z["tri.in_crc"]
    got = fn_ctx.eval_fn("tri", verts, normals, o, d)
    assert count_diff(got[:, 0].astype(np.uint8), z["tri.hit"]) == 0
    assert bit_equal(got[:, 1], z["tri.t"])
    assert bit_equal(got[:, 2:5], z["tri.pos"]) and bit_equal(got[:, 5:8], z["tri.Ns"]) and bit_equal(got[:, 8:11], z["tri.Ng"])
    assert 0.05 < z["tri.hit"].mean() < 0.95


def test_device_math_helpers(fn_ctx, port):
    """global.hpp helpers on the device against the reference build: exact for + - * / sqrt chains, <= 16 ulp or 1e-6 + 2e-5
    relative through acosf / tanf / x^5"""
    z = np.load(golden_path("functions.npz"))
    a, b, c, scale, eta_i, eta_t, rough, x, y = pc.math_inputs()
    assert pc.checksum(a, b, c, scale, eta_i, eta_t, rough, x, y) == z["math.in_crc"]
    h = port.normalized(a + b)  # an input of D / G (as parity_cases.run_math forms it)
    cos = np.clip((a * b).sum(1), -1, 1).astype(np.float32)
    F0 = np.abs(c).astype(np.float32)
    zero = np.zeros((4, 3), np.float32)
    f = fn_ctx.eval_fn
    exact = {
        "normalized": f("normalized", np.concatenate([a * scale, zero])),
        "reflect": f("reflect", a * scale, b),
        "refract": f("refract", a, b * scale, eta_i, eta_t),
        "D": f("D", h, b, rough)[:, 0],
        "mis": f("mis", x, y)[:, 0],
        "local2world": f("local2world", a * scale, b),
    }
    for k, g in exact.items():
        assert count_diff(g, z[f"math.{k}"]) == 0, k
    # (until round 5 these three went through the device's own sinf / acosf / tanf / x^5 and were held to 16 ulp; with the C
    # library's functions restated in csrc/device_libm.h they are the reference build's bits)
    libm = {
        "fresnel": f("fresnel", a * scale, b, eta_i, eta_t)[:, 0],
        "fresnel_schlick": f("fresnel_schlick", cos, F0),
        "G": f("G", a, c, b, rough, h)[:, 0],
    }
    for k, g in libm.items():
        assert count_diff(g, z[f"math.{k}"]) == 0, k


def test_device_libm_is_the_c_librarys(fn_ctx):
    """csrc/device_libm.h on the DEVICE against the C library of this box (glibc 2.35, the one the reference build links): sinf,
    cosf, acosf, tanf, powf, atanf, atan2f on random bit patterns of the whole float32 range, on the ranges the path calls them with, and on the
    special values.  (Exhaustively, compiled for the host: tests/test_libm_restatement.py and tests/tools/libm_check.c.)"""
    import ctypes

    ver = ctypes.CDLL(None).gnu_get_libc_version
    ver.restype = ctypes.c_char_p
    if ver().decode() != "2.35":
        pytest.skip("device_libm.h restates glibc 2.35")
    m = ctypes.CDLL("libm.so.6")
    for nm in ("sinf", "cosf", "acosf", "tanf", "atanf"):
        getattr(m, nm).restype = ctypes.c_float
        getattr(m, nm).argtypes = [ctypes.c_float]
    for nm in ("powf", "atan2f"):
        getattr(m, nm).restype = ctypes.c_float
        getattr(m, nm).argtypes = [ctypes.c_float, ctypes.c_float]
    rng = np.random.default_rng(20261005)
    n = 20000
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1.17549435e-38, 3.4028235e38, 0.5, 2.0, 5.0,
                        np.pi, np.pi / 2, np.pi / 4, 2 * np.pi, 0.785398185, 1.57079637, 3.14159274, 6.28318548, 1e9, 1e20, 1e30], np.float32)
    x = np.concatenate([
        rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32),  # any bit pattern
        (rng.random(n) * 2 * np.pi).astype(np.float32),  # phi = 2 pi xi (sampleDirection)
        (rng.random(n) * 2 - 1).astype(np.float32),  # cosines (acosf in G_smf, the sphere texture)
        (rng.random(n) * np.pi).astype(np.float32),  # angles (tanf in G_smf)
        (1 - rng.random(n) ** 4).astype(np.float32),  # near 1: grazing
        special,
    ])
    y = np.concatenate([np.full(n, 5.0), np.full(n, 2.0), rng.normal(0, 3, n), np.full(n, -1.5), np.full(n, 0.5), special[::-1]]).astype(np.float32)
    got = fn_ctx.eval_fn("libm", x, y)
    want = np.empty_like(got)
    for i, (xi, yi) in enumerate(zip(x.tolist(), y.tolist())):
        want[i] = (m.sinf(xi), m.cosf(xi), m.acosf(xi), m.tanf(xi), m.powf(xi, yi), m.powf(xi, 5.0), m.atan2f(xi, yi), m.atanf(xi))
    for k, nm in enumerate(("sinf", "cosf", "acosf", "tanf", "powf(x, y)", "powf(x, 5)", "atan2f(x, y)", "atanf")):
        g, w = got[:, k], want[:, k]
        bad = (g.view(np.uint32) != w.view(np.uint32)) & ~(np.isnan(g) & np.isnan(w))
        assert not bad.any(), (nm, int(bad.sum()), x[bad][:4], y[bad][:4], g[bad][:4], w[bad][:4])


def test_device_rng_known_answers(fn_ctx, port):
    """Philox4x32-10 on the device: the Random123 known-answer vector with a zero last counter word (the only kind this path
    forms), 300 random (pixel, sample, block, key) blocks against the CPU implementation, and the xi stream of one sample
    against the reference build's golden stream"""
    z = np.load(golden_path("functions.npz"))
    q = np.array([[0, 0, 0, 0, 0]], np.uint32)
    out = fn_ctx.eval_fn("philox", q).view(np.uint32)
    assert [hex(v) for v in out[0]] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert bit_equal(out[0], z["philox.out"][0])
    r = np.random.default_rng(77)
    q = r.integers(0, 2 ** 32, (300, 5), dtype=np.uint64).astype(np.uint32)
    q[:, 2] &= 0x3FFFFFFF  # block = draw >> 2
    got = fn_ctx.eval_fn("philox", q).view(np.uint32)
    # the CPU implementation takes one key per call: check 300 rows one by one
    for i in range(300):
        ctr = np.array([[q[i, 0], q[i, 1], q[i, 2], 0]], np.uint32)
        assert bit_equal(port.philox(ctr, int(q[i, 3]), int(q[i, 4]))[0], got[i]), i
    # the xi stream: draws 0..63 of sample 7 of pixel 12345 under key (KEY0, 2)
    rows = np.array([[12345, 7, pc.KEY0, 2, 8 * k] for k in range(8)], np.uint32)
    xi = fn_ctx.eval_fn("rng", rows).reshape(-1)
    assert bit_equal(xi, z["philox.stream"])
    assert xi.min() >= 0 and xi.max() < 1


# ------------------------------------------------------------------------------------------------------------
# the BASELINE frames themselves
def _frame_checks(tr, port, sc, key1, spp, n_probe_pixels=12, n_samples=2000, low_spp=2):
    """Properties of a full-size, full-spp frame that do not need a CPU render of it:
      * the frame with the default four passes in flight == the frame with one pass in flight, bit for bit;
      * a pixel of the frame == (1/spp) * the sum, in sample order and dropping NaN samples, of that pixel's per-sample
        radiances (tutu_hip_trace_samples) -- the resolve of PathTracing.hpp:507-513, bit for bit;
      * 2000 random (pixel, sample) pairs against the CPU restatement at the per-sample bar;
      * the frame's mean against a low-spp CPU render of the same frame within Monte-Carlo error."""
    W, H = sc["width"], sc["height"]
    with tr.Context(sc) as ctx:
        frame = ctx.render(spp, pc.KEY0, key1)                       # host-pointer path (tutu_hip_render), default work sets
        st = ctx.last_stats
        assert st["samples"] == W * H * spp and st["passes"] > 4 and st["n_sets"] == 4
        ctx.set_option("sets", 1)
        one = ctx.render(spp, pc.KEY0, key1)
        assert ctx.last_stats["n_sets"] == 1
        ctx.set_option("sets", 0)
        assert bit_equal(frame, one)
        assert np.isfinite(frame).all()
        rng = np.random.default_rng(2024)
        probe = rng.integers(0, W * H, n_probe_pixels)
        pix = np.repeat(probe.astype(np.uint32), spp)
        smp = np.tile(np.arange(spp, dtype=np.uint32), n_probe_pixels)
        L = ctx.trace_samples(pix, smp, pc.KEY0, key1).reshape(n_probe_pixels, spp, 3)
        acc = np.zeros((n_probe_pixels, 3), np.float32)
        for s in range(spp):
            ok = ~np.isnan(L[:, s]).any(1)
            acc[ok] = acc[ok] + L[ok, s]
        want_px = acc * np.float32(1.0 / spp)
        assert bit_equal(frame.reshape(-1, 3)[probe], want_px)
        pix = rng.integers(0, W * H, n_samples).astype(np.uint32)
        smp = rng.integers(0, spp, n_samples).astype(np.uint32)
        Lg = ctx.trace_samples(pix, smp, pc.KEY0, key1)
        ctx.set_option("exact_sum", 1)
        Lx = ctx.trace_samples(pix, smp, pc.KEY0, key1)
        ctx.set_option("exact_sum", 0)
    S = port.scene(sc)
    Lw = S.trace_samples(pix, smp, pc.KEY0, key1)
    low = S.render(low_spp, pc.KEY0, key1, nthreads=16)
    S.close()
    fin = ~(np.isnan(Lg).any(1) | np.isnan(Lw).any(1))
    err = np.abs(Lg[fin] - Lw[fin]).max(1)
    scale = np.maximum(np.abs(Lw[fin]).max(1), 1e-3)
    assert (err <= 5e-6 * scale + 1e-7).all(), float((err / scale).max())  # (every sample takes the reference's path: module docstring)
    assert not ((Lx.view(np.uint32) != Lw.view(np.uint32)) & ~(np.isnan(Lx) & np.isnan(Lw))).any()  # knob exact_sum: the CPU restatement's bits
    # Monte-Carlo error of the low-spp CPU mean: per-channel variance of its pixels / number of pixels (plus the frame's own, smaller)
    for ch in range(3):
        sigma = low[..., ch].std() / np.sqrt(low[..., ch].size) * 1.5
        assert abs(frame[..., ch].mean() - low[..., ch].mean()) < 5 * sigma + 1e-4, (ch, frame[..., ch].mean(), low[..., ch].mean(), sigma)
    return frame, st


def test_headline_frame_cornell_800x800_512spp(tr, port):
    """BASELINE configs[1], the frame bench.py times: 27 passes x 4 streams x 12 Mi record slots"""
    from tuturenderer_amd import scenes

    frame, st = _frame_checks(tr, port, scenes.cornell_box(800, 800), key1=2, spp=512)
    assert abs(frame.mean() - 0.3960) < 0.004  # the reference's own mean at 800x800
    print(f"cornell 800x800x512: mean {frame.mean():.5f} passes {st['passes']} spp/pass {st['spp_per_pass']} rays/sample "
          f"{(st['closest_rays'] + st['shadow_rays']) / st['samples']:.2f}")


def test_config5_frame_veach_800x600_512spp(tr, port):
    """BASELINE configs[4] at its size and spp (mixed materials: Lambertian, PERFECT_REFRACTIVE, MICROFACET_R -> SHADE_ANY)"""
    from tuturenderer_amd import scenes

    frame, st = _frame_checks(tr, port, scenes.veach_room(800, 600, small_light=False), key1=5, spp=512, n_probe_pixels=8)
    assert abs(frame.mean() - 0.2149) < 0.01  # veach without the small light


def test_config3_frame_bunny_stand_in_1024x1024_256spp(tr, port):
    """BASELINE configs[2] at its size and spp on the stand-in mesh (81 942 triangles, MICROFACET_T blob: tree in HBM)"""
    from tuturenderer_amd import scenes

    frame, st = _frame_checks(tr, port, scenes.bunny_box(1024, 1024), key1=3, spp=256, n_probe_pixels=8)
    assert 0.2 < frame.mean() < 0.6 and st["passes"] >= 8


def test_config4_frame_broom_stand_in_1600x900_1024spp(tr, port):
    """BASELINE configs[3] at its size and spp on the stand-in (48 012 thin prisms, 1.37 M references in the walked tree):
    1.47 G samples per frame, twice (four passes in flight == one)"""
    from tuturenderer_amd import scenes

    frame, st = _frame_checks(tr, port, scenes.broom_room(1600, 900), key1=4, spp=1024, n_probe_pixels=4, n_samples=1500, low_spp=1)
    assert st["samples"] == 1600 * 900 * 1024 and frame.mean() > 0.01


def test_two_ranks_on_one_gpu_gather_the_single_process_frame(tr, tmp_path):
    """The N > 1 path with REAL HIP contexts: two processes (gloo rendezvous, both on cuda:0) each render their pixel
    tiles with their own TutuCtx and FrameGather assembles the frame on rank 0 -- bit-identical to one process rendering
    everything (the RNG is keyed by global pixel and sample).  The RCCL leg of the same code is bench.py --gpus N."""
    import subprocess
    import sys

    from conftest import ROOT
    from tuturenderer_amd import scenes

    out = str(tmp_path / "frame2.npy")
    port_no = 29500 + (hash(out) % 400)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port_no), "bench.py", "--gpus", "2", "--same-device", "--backend", "gloo", "--config", "c1", "--spp", "6",
           "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--dump", out]
    r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    import json

    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    # the line proves that the collective library saw both ranks (an all-reduce of 1) and carries every rank's own time
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["collective_backend"] == "gloo" and len(line["rank_ms_per_step"]) == 2
    two = np.load(out)
    with tr.Context(scenes.cornell_box(800, 800)) as ctx:
        one = ctx.render(6, pc.KEY0, 1)
    assert bit_equal(two, one)


def test_rccl_process_group_self_loop_gathers_the_plain_frame(tr, tmp_path):
    """The `nccl` (= RCCL) leg of the N > 1 path, executed on the hardware there is: ONE rank under torch.distributed.run, the
    process group really initialised with backend nccl, and every frame's piece sent through ONE grouped self send / recv
    (FrameGather(self_loop=True): the same dist.batch_isend_irecv call as with N ranks) before the un-tiling kernel -- bit-equal
    to the plain call.  (What this cannot show: more than one device, xGMI.  PathTracing.hpp:393-429 is what the gather replaces.)"""
    import os
    import subprocess
    import sys

    from conftest import ROOT
    from tuturenderer_amd import scenes

    out = str(tmp_path / "frame_loop.npy")
    port_no = 29900 + (hash(out) % 90)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port_no), "bench.py", "--gpus", "1", "--self-loop", "--backend", "nccl", "--config", "c1", "--spp", "6",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--dump", out]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    import json

    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["rccl_ranks"] == 1 and line["collective_backend"] == "nccl" and len(line["rank_ms_per_step"]) == 1
    assert line["config"]["gpu_busy_probe"]["samples"] >= 0
    looped = np.load(out)
    with tr.Context(scenes.cornell_box(800, 800)) as ctx:
        one = ctx.render(6, pc.KEY0, 1)
    assert bit_equal(looped, one)


def test_cold_start_first_render_is_not_an_allocation(tr):
    """The reference's scene programs render once per process (src/main_cornellBox.cpp:75-79): create + FIRST render is what a
    drop-in user waits for.  Round 3 allocated 67 GB of work sets before tracing the 32-triangle box (first render 1.84 s for
    a 0.12 s frame).  Now the first default-sized render allocates `cold_paths_mi` Mi path slots itself, a host thread brings
    the full-size sets, and a later render adopts them: create + first render stay within 1 s (north_star's 50 x the
    reference CPU = 1.0 s for this frame) and within 2 x a steady-state render + 0.1 s; every render returns the same bits."""
    import os
    import time

    from tuturenderer_amd import scenes

    sc = scenes.cornell_box(800, 800)
    t0 = time.perf_counter()
    ctx = tr.Context(sc)
    try:
        first = ctx.render(512, pc.KEY0, 2, full_frame=False)
        t_cold = time.perf_counter() - t0
        cold_paths = ctx.get_option("work_paths_mi")
        t0 = time.perf_counter()
        second = ctx.render(512, pc.KEY0, 2, full_frame=False)
        t_second = time.perf_counter() - t0
        ctx.work_ready(wait=True)
        t0 = time.perf_counter()
        third = ctx.render(512, pc.KEY0, 2, full_frame=False)
        t_steady = time.perf_counter() - t0
        steady_paths = ctx.get_option("work_paths_mi")
        grow_ms = ctx.get_option("grow_ms")
    finally:
        ctx.close()
    print(f"\n[cold start] create + first render {t_cold:.3f} s on {cold_paths} Mi path slots; second {t_second:.3f} s; steady {t_steady:.3f} s on "
          f"{steady_paths} Mi (background allocation {grow_ms} ms)")
    assert bit_equal(first, second) and bit_equal(first, third)
    assert cold_paths <= 16 and steady_paths >= 96
    # wall-clock bounds hold on an idle box (0.15 s / 0.12 s measured) but the cost of a device allocation varies with the state
    # of the driver's memory and the host may be busy: printed always, asserted only on request (TUTU_TEST_TIMING=1)
    if os.environ.get("TUTU_TEST_TIMING") == "1":
        assert t_cold < 1.0, t_cold
        assert t_cold < 2 * t_steady + 0.1, (t_cold, t_steady)
