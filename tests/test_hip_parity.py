"""GPU (MI355X): the HIP path, called through the C ABI, against the CPU oracle and the golden vectors that the
reference build produced.

Bars:
  * traversal decisions (hit / triangle index / blocked) and every +,-,*,/,sqrt-only quantity (t, barycentrics,
    light samples, camera rays): BIT-EXACT;
  * material functions that go through sinf/cosf/acosf/tanf/pow: <= 16 ulp of float32, or 1e-6 absolute + 2e-5
    relative (the acosf->tanf chain of G_smf is ill-conditioned near grazing angles);
  * per-sample radiance at matched Philox seed: relative 1e-4 for >= 99.5 % of the samples (the rest are discrete
    branch flips caused by a last-bit difference in a transcendental);
  * images at matched seed: mean per-pixel L2 < 1e-3 (the tolerance BASELINE.json's north_star states).
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
    for k in ("s_ok", "s_special", "s_ndraws"):
        assert count_diff(got[k], want[k]) <= 2, k  # discrete outcomes (a flip needs xi within an ulp of F)
    for k in ("bxdf", "bxdf_tir", "pdf", "s_wi", "s_pdf", "s_bxdf"):
        g, w = np.asarray(got[k], np.float32), np.asarray(want[k], np.float32)
        close = (ulp_diff(g, w) <= 16) | (np.abs(g - w) <= 1e-6 + 2e-5 * np.abs(w))
        # FLOAT_EQUAL / `< 0` branch flips on last-bit input differences are allowed to be rare
        assert (~close).mean() < 2e-3, (k, int((~close).sum()), g[~close][:4], w[~close][:4])


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
    ok = err <= 1e-4 * scale + 1e-6
    frac_bad = 1.0 - ok.mean()
    print(f"{name}: NaN samples {int(nan_w.sum())}; diverged {int((~ok).sum())}/{len(ok)} max rel err of the rest {float((err[ok] / scale[ok]).max()):.2e}")
    # A shadow ray towards a point sampled on a sphere light ends 5e-4 in front of that sphere, and the sphere's own
    # float quadratic (error ~3e-4 at room-scale distances) decides whether the light blocks itself: the reference's
    # answer there is rounding noise, and a last-bit difference upstream flips it.  1 % instead of 0.5 % for that scene.
    assert frac_bad < (1e-2 if name == "cornell_spheres" else 5e-3), frac_bad
    assert abs(L[fin].mean() - want[fin].mean()) < 2e-2 * max(want[fin].mean(), 1e-3)
    S.close()


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
    assert l2.mean() < 1e-3
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


@pytest.mark.parametrize("name,n", [("bunny", 600), ("broom", 300)])
def test_stand_in_scenes_per_sample_parity(tr, port, name, n):
    """BASELINE configs 3 and 4 (synthetic stand-ins, 82 k / 48 k triangles, BVH depth 17 / 16, rough glass /
    rough metal): these go through the HBM-resident traversal path.  Matched-seed radiance of random samples
    against the CPU restatement."""
    from tuturenderer_amd import scenes

    sc = scenes.bunny_box(256, 256) if name == "bunny" else scenes.broom_room(320, 180)
    key1 = 3 if name == "bunny" else 4
    rng = np.random.default_rng(5)
    pix = rng.integers(0, sc["width"] * sc["height"], n).astype(np.uint32)
    smp = rng.integers(0, 64, n).astype(np.uint32)
    with tr.Context(sc) as ctx:
        info = ctx.info()
        L = ctx.trace_samples(pix, smp, pc.KEY0, key1)
        # kernel-level: the same primary rays through the closest-hit entry point, bit-exact against the oracle
        S = port.scene(sc)
        d = S.raydir((pix % sc["width"]).astype(np.int32), (pix // sc["width"]).astype(np.int32))
        o = np.repeat(S.camera()[5][None], n, 0)
        hits = ctx.trace_closest(o, d)
    hit, t, tri, *_ = S.closest(o, d)
    assert count_diff(hits["tri"], tri) == 0 and count_diff(np.where(tri >= 0, hits["t"], 0), np.where(tri >= 0, t, 0)) == 0
    want = S.trace_samples(pix, smp, pc.KEY0, key1)
    S.close()
    err = np.abs(L - want).max(1)
    scale = np.maximum(np.abs(want).max(1), 1e-3)
    bad = ~((err <= 1e-4 * scale + 1e-6) | (np.isnan(L).any(1) & np.isnan(want).any(1)))
    print(f"{name}: {info} diverged {int(bad.sum())}/{n}")
    assert info["depth"] >= 15
    assert bad.mean() < 0.02
