"""TEST INFRASTRUCTURE ONLY -- seeded input generators + case runners shared by oracle/gen_golden.py (which runs
them on the REFERENCE build and stores the outputs under tests/golden/) and by tests/ (which run them on the port
and on the HIP library and compare).

Every runner takes an object with the oracle.pyoracle.Oracle / OracleScene method names and returns a dict of
numpy arrays.  Inputs are regenerated from the seed, so the golden files hold outputs only (plus a checksum of the
inputs, to catch generator drift).
"""
import zlib

import numpy as np

KEY0 = 0x5EED0001  # Philox key word 0


def _rng(seed):
    return np.random.Generator(np.random.Philox(key=seed))


def unit(rng, n):
    v = rng.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return v.astype(np.float32)


def xi24(rng, shape):
    """uniform 24-bit floats in [0,1) -- exactly representable, so injection is lossless"""
    return (rng.integers(0, 1 << 24, size=shape).astype(np.float32) / np.float32(1 << 24)).astype(np.float32)


def checksum(*arrs):
    c = 0
    for a in arrs:
        c = zlib.crc32(np.ascontiguousarray(a).tobytes(), c)
    return np.uint32(c)


# ------------------------------------------------------------------------------------------------ pure functions
def bbox_inputs(n=20000, seed=101):
    r = _rng(seed)
    a = r.uniform(-10, 10, (n, 3)).astype(np.float32)
    b = r.uniform(-10, 10, (n, 3)).astype(np.float32)
    pmin = np.minimum(a, b)
    pmax = np.maximum(a, b)
    # flat boxes (axis-aligned walls) and degenerate points
    flat = r.integers(0, 4, n)
    for ax in range(3):
        m = flat == ax
        pmax[m, ax] = pmin[m, ax]
    o = r.uniform(-15, 15, (n, 3)).astype(np.float32)
    d = unit(r, n)
    # two thirds of the rays are aimed at a point of the box (interior, face or corner), so hits are common
    w = r.random((n, 3))
    corner = r.integers(0, 4, n) == 0
    w[corner] = np.round(w[corner])
    tgt = pmin + (pmax - pmin) * w
    aim = r.random(n) < 0.67
    da = tgt - o
    da /= np.maximum(np.linalg.norm(da, axis=1, keepdims=True), 1e-20)
    d[aim] = da[aim].astype(np.float32)
    inside = r.random(n) < 0.05
    o[inside] = tgt[inside].astype(np.float32)
    # zero direction components (+0 and -0), origins on slab planes
    z = r.integers(0, 8, n)
    for ax in range(3):
        d[z == ax, ax] = 0.0
        d[z == ax + 3, ax] = -0.0
    on = r.integers(0, 6, n)
    for ax in range(3):
        m = on == ax
        o[m, ax] = pmin[m, ax]
    return pmin, pmax, o, d


def run_bbox(O):
    pmin, pmax, o, d = bbox_inputs()
    return {"hit": O.bbox_intersect(pmin, pmax, o, d), "in_crc": checksum(pmin, pmax, o, d)}


def tri_inputs(n=20000, seed=102):
    r = _rng(seed)
    verts = r.uniform(-5, 5, (n, 9)).astype(np.float32)
    normals = unit(r, 3 * n).reshape(n, 9)
    # aim most rays at the triangle (interior, edges, vertices), rest random
    w = r.dirichlet((1, 1, 1), n).astype(np.float32)
    kind = r.integers(0, 6, n)
    w[kind == 0] = np.array([0.5, 0.5, 0.0], np.float32)  # on an edge
    w[kind == 1] = np.array([1.0, 0.0, 0.0], np.float32)  # on a vertex
    v = verts.reshape(n, 3, 3)
    target = (v * w[:, :, None]).sum(1)
    o = r.uniform(-8, 8, (n, 3)).astype(np.float32)
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    # grazing rays: direction (almost) in the triangle plane
    g = kind == 2
    e1 = v[:, 1] - v[:, 0]
    e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
    nrm = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    eps = r.uniform(-3e-4, 3e-4, n)[:, None]
    dg = e1 + eps * nrm
    dg /= np.linalg.norm(dg, axis=1, keepdims=True)
    d[g] = dg[g].astype(np.float32)
    rnd = kind == 3
    d[rnd] = unit(r, int(rnd.sum()))
    return verts, normals, o.astype(np.float32), d


def run_tri(O):
    verts, normals, o, d = tri_inputs()
    hit, t, pos, Ns, Ng = O.tri_intersect(verts, normals, o, d)
    h = hit.astype(bool)
    return {"hit": hit, "t": np.where(h, t, 0).astype(np.float32), "pos": pos * h[:, None], "Ns": Ns * h[:, None],
            "Ng": Ng * h[:, None], "area": O.tri_area(verts), "in_crc": checksum(verts, normals, o, d)}


def math_inputs(n=10000, seed=103):
    r = _rng(seed)
    a, b, c = unit(r, n), unit(r, n), unit(r, n)
    scale = r.uniform(0.1, 3.0, (n, 1)).astype(np.float32)
    eta_i = np.where(r.random(n) < 0.5, 1.0, 1.5).astype(np.float32)
    eta_t = np.where(eta_i == 1.0, 1.5, 1.0).astype(np.float32)
    rough = r.uniform(0.0, 1.0, n).astype(np.float32)
    rough[:50] = 0.0
    x = r.uniform(0, 5, n).astype(np.float32)
    y = r.uniform(0, 5, n).astype(np.float32)
    return a, b, c, scale, eta_i, eta_t, rough, x, y


def run_math(O):
    a, b, c, scale, eta_i, eta_t, rough, x, y = math_inputs()
    h = O.normalized(a + b)
    cos = np.clip((a * b).sum(1), -1, 1).astype(np.float32)
    F0 = np.abs(c).astype(np.float32)
    zero = np.zeros((4, 3), np.float32)
    return {
        "normalized": O.normalized(np.concatenate([a * scale, zero])),
        "fresnel": O.fresnel(a * scale, b, eta_i, eta_t),
        "fresnel_schlick": O.fresnel_schlick(cos, F0),
        "reflect": O.reflect(a * scale, b),
        "refract": O.refract(a, b * scale, eta_i, eta_t),
        "D": O.D(h, b, rough),
        "G": O.G(a, c, b, rough, h),
        "mis": O.mis(x, y),
        "local2world": O.local2world(a * scale, b),
        "write_pixel": O.write_pixel(np.linspace(-0.1, 1.2, 4096, dtype=np.float32)),
        "in_crc": checksum(a, b, c, scale, eta_i, eta_t, rough, x, y),
    }


def material_set():
    from oracle.pyoracle import (LAMBERTIAN, MICROFACET_R, MICROFACET_T, PERFECT_REFLECTIVE, PERFECT_REFRACTIVE, UNLIT,
                                 make_material)
    return [
        ("lambert", make_material(LAMBERTIAN, (0.725, 0.71, 0.68))),
        ("mirror", make_material(PERFECT_REFLECTIVE)),
        ("glass", make_material(PERFECT_REFRACTIVE, eta=1.5)),
        ("ggx_r", make_material(MICROFACET_R, (0.3296, 0.2580, 0.1503), roughness=0.2775146484375, metallic=0.5)),
        ("ggx_r_rough", make_material(MICROFACET_R, (0.8, 0.5, 0.2), roughness=0.8, metallic=0.0, alpha=0.5)),
        ("ggx_t", make_material(MICROFACET_T, (0.9, 0.9, 0.9), eta=1.5, roughness=0.2)),
        ("ggx_t_rough", make_material(MICROFACET_T, (0.9, 0.9, 0.9), eta=1.33, roughness=0.6)),
        ("unlit", make_material(UNLIT, (0.2, 0.4, 0.6))),
    ]


def material_inputs(n=6000, seed=104):
    r = _rng(seed)
    Ng = unit(r, n)
    # shading normal: a perturbed geometric normal (what smooth meshes give), sometimes identical
    Ns = Ng + 0.3 * r.normal(size=(n, 3)).astype(np.float32)
    Ns /= np.linalg.norm(Ns, axis=1, keepdims=True)
    same = r.random(n) < 0.4
    Ns[same] = Ng[same]
    Ns = Ns.astype(np.float32)
    wo = unit(r, n)
    wi = unit(r, n)
    tir = (r.random(n) < 0.1).astype(np.uint8)
    xi = xi24(r, (n, 3))
    return Ng, Ns, wo, wi, tir, xi


def run_material(O, name, mat, s_wi_at=None):
    Ng, Ns, wo, wi, tir, xi = material_inputs()
    n = len(Ng)
    out = {}
    # exact mirror / refraction directions for the delta lobes, so the FLOAT_EQUAL branches are exercised
    wi2 = wi.copy()
    refl = O.normalized(O.reflect(wo, Ns))
    refr = O.normalized(O.refract(wo, Ns, np.ones(n, np.float32), np.full(n, float(mat["eta"]), np.float32)))
    wi2[0::3] = refl[0::3]
    nz = (np.abs(refr).sum(1) > 0)
    sel = np.zeros(n, bool)
    sel[1::3] = True
    wi2[sel & nz] = refr[sel & nz]
    for eta_scene in (1.0,):
        out["bxdf"] = O.mat_bxdf(mat, wi2, wo, Ng, Ns, eta_scene, tir=None)
        out["bxdf_tir"] = O.mat_bxdf(mat, wi2, wo, Ng, Ns, eta_scene, tir=tir)
        out["pdf"] = O.mat_pdf(mat, wi2, wo, Ns, eta_scene, float(mat["eta"]))
    swi, ok, sp, nd = O.mat_sample(mat, wo, Ns, xi, 1.0)
    valid = (ok.astype(bool) & ~sp.astype(bool))[:, None]
    out["s_wi"] = np.where(valid, swi, 0).astype(np.float32)
    out["s_ok"] = ok
    out["s_special"] = sp
    out["s_ndraws"] = nd
    # pdf / bxdf at the sampled direction (what the integrator evaluates)
    # (a device under test evaluates them at the GOLDEN sampled direction, s_wi_at, so that the comparison sees the
    # function's own error and not the GGX lobe's huge sensitivity to a last-bit difference in its input)
    swi_n = O.normalized(out["s_wi"] if s_wi_at is None else s_wi_at)
    out["s_pdf"] = O.mat_pdf(mat, swi_n, wo, Ns, 1.0, float(mat["eta"]))
    out["s_bxdf"] = O.mat_bxdf(mat, swi_n, wo, Ng, Ns, 1.0, tir=None)
    out["in_crc"] = checksum(Ng, Ns, wo, wi, tir, xi)
    return {f"{name}.{k}": v for k, v in out.items()}


# ------------------------------------------------------------------------------------------------ scene level
def scene_rays(S, n_cam=4000, n_rand=6000, seed=105):
    """camera rays through random pixels + random rays from inside the scene's bounding box"""
    r = _rng(seed)
    px = r.integers(0, S.W, n_cam).astype(np.int32)
    py = r.integers(0, S.H, n_cam).astype(np.int32)
    dcam = S.raydir(px, py)
    eye = S.camera()[5]
    lo = S.verts.reshape(-1, 3).min(0)
    hi = S.verts.reshape(-1, 3).max(0)
    o = (lo + (hi - lo) * r.random((n_rand, 3))).astype(np.float32)
    d = unit(r, n_rand)
    # axis-parallel directions with exact zeros: NaN/inf slab cases on flat boxes
    k = n_rand // 10
    ax = r.integers(0, 3, k)
    dz = np.zeros((k, 3), np.float32)
    dz[np.arange(k), ax] = np.where(r.random(k) < 0.5, 1.0, -1.0)
    d[:k] = dz
    O = np.concatenate([np.repeat(eye[None], n_cam, 0), o]).astype(np.float32)
    D = np.concatenate([dcam, d]).astype(np.float32)
    return O, D


def run_scene(S, seed=105):
    O, D = scene_rays(S, seed=seed)
    hit, t, tri, pos, Ns, Ng = S.closest(O, D)
    h = hit.astype(bool)
    out = {"hit": hit, "t": np.where(h, t, 0).astype(np.float32), "tri": tri, "pos": pos * h[:, None],
           "Ns": Ns * h[:, None], "Ng": Ng * h[:, None]}
    # shadow queries: from hit points (offset like the integrator) to points sampled on the lights, plus
    # targets exactly on geometry so the `t < dist && !FLOAT_EQUAL` rule is exercised
    r = _rng(seed + 1)
    idx = np.nonzero(h)[0]
    xi = xi24(r, (len(idx), 3))
    ltri, lpos, lnrm, lpdf = S.sample_light(xi)
    orig = (pos[idx] + np.float32(0.0005) * Ns[idx] * np.sign((Ns[idx] * -D[idx]).sum(1, keepdims=True))).astype(np.float32)
    target = (lpos + np.float32(0.0005) * lnrm).astype(np.float32)
    half = len(idx) // 2
    target[half:] = pos[idx][::-1][half:]  # geometry-to-geometry
    out["blocked"] = S.any_hit(orig, target)
    out["light_tri"], out["light_pos"], out["light_nrm"], out["light_pdf"] = ltri, lpos, lnrm, lpdf
    lights = S.lights()
    out["lights"] = lights
    out["light_pdf_of_tri"] = S.light_pdf(np.arange(len(S.verts), dtype=np.int32))
    out["camera"] = S.camera()
    out["in_crc"] = checksum(O, D, xi)
    return out


def sample_ids(S, n=3000, seed=106, spp=4):
    r = _rng(seed)
    pix = r.integers(0, S.W * S.H, n).astype(np.uint32)
    pix = np.repeat(pix, spp)
    smp = np.tile(np.arange(spp, dtype=np.uint32), n)
    return pix, smp


def run_samples(S, key1, n=3000, spp=4):
    pix, smp = sample_ids(S, n=n, spp=spp)
    L, nd, nc = S.trace_samples(pix, smp, KEY0, key1, stats=True)
    return {"L": L, "ndraws": nd, "nclosest": nc, "in_crc": checksum(pix, smp)}


def degenerate_cornell(W=64, H=64):
    """Edge-case scene: the Cornell box with a floor triangle whose vertex normals are all zero (Ns = 0), a ceiling
    triangle with one zero vertex normal, a zero-area (collinear) diffuse triangle and a ZERO-AREA LIGHT.  The light's
    pdf 1/(n*0) is +inf, the power heuristic becomes inf/inf, and about one sample in six comes out NaN -- which
    sub_render_pt drops while still dividing by SPP (PathTracing.hpp:510-513)."""
    from tuturenderer_amd import scenes

    sc = scenes.cornell_box(W, H)
    v, n, m = sc["verts"].copy(), sc["normals"].copy(), sc["mat_id"].copy()
    n[0] = 0
    n[3, :3] = 0
    extra_v = np.array([[100, 100, 100, 200, 200, 200, 300, 300, 300], [260, 300, 200, 270, 300, 200, 280, 300, 200]], np.float32)
    extra_n = np.array([[0, 1, 0] * 3, [0, -1, 0] * 3], np.float32)
    sc["verts"] = np.concatenate([v, extra_v])
    sc["normals"] = np.concatenate([n, extra_n])
    sc["mat_id"] = np.concatenate([m, np.array([0, 1], np.int32)])
    return sc


def texture_inputs(n=4000, seed=31):
    """(u, v) for Texture::getRGBat: a spread over [-3,3] plus the exact integers and the wrap edge cases."""
    r = np.random.default_rng(seed)
    u = (r.random(n, dtype=np.float32) * 6 - 3).astype(np.float32)
    v = (r.random(n, dtype=np.float32) * 6 - 3).astype(np.float32)
    edge = np.array([0.0, 1.0, -1.0, 2.0, -2.0, 0.5, -0.5, 0.99999994, -0.99999994, 1e-8, -1e-8, 3.0], np.float32)
    u[:len(edge)] = edge
    v[:len(edge)] = edge[::-1]
    u[len(edge):2 * len(edge)] = edge
    v[len(edge):2 * len(edge)] = 0.25
    return u, v


def run_texture(O, img):
    u, v = texture_inputs()
    return {"in_crc": checksum(u, v, np.ascontiguousarray(img, np.float32)), "rgb": O.texture_lookup(img, u, v)}


def postprocess_input(h=40, w=56, seed=41):
    """an HDR image with a few pixels above the bloom threshold (|c| > 3) and a bright patch"""
    r = np.random.default_rng(seed)
    img = (r.random((h, w, 3), dtype=np.float32) ** 4 * np.float32(6)).astype(np.float32)
    img[10:14, 20:26] = [9, 7, 5]
    img[0, 0] = [4, 4, 4]
    img[h - 1, w - 1] = [0.5, 8, 0.25]
    return img


def run_postprocess(O):
    img = postprocess_input()
    out = {"in_crc": checksum(img)}
    for stage, name in ((0, "full"), (1, "emissive"), (2, "blur"), (3, "hdr")):
        out[name] = O.postprocess(stage, img)
    return out


# ------------------------------------------------------------------------------------------------ adversarial geometry
def needle_scene(n=2000, length=1000.0, width=1.0, seed=5, box=1000.0):
    """2 n needle-shaped triangles (n quads of length : width = 1000 : 1, random orientation) -- the geometry on which the
    reference's fp32 triangle test (Triangle.hpp:23-59) is ill-conditioned: half the triangles have an angle of ~1e-3 rad at
    v0.  Returned as a scene dict on the Cornell camera (only the ray-batch entry points are used on it)."""
    from tuturenderer_amd import scenes

    r = np.random.default_rng(seed)
    c = r.uniform(0.2 * box, 0.8 * box, (n, 3))
    ax = r.normal(size=(n, 3))
    ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    sd = np.cross(ax, r.normal(size=(n, 3)))
    sd /= np.linalg.norm(sd, axis=1, keepdims=True)
    p0 = c - ax * length / 2 - sd * width / 2
    p1 = c - ax * length / 2 + sd * width / 2
    p2 = c + ax * length / 2 + sd * width / 2
    p3 = c + ax * length / 2 - sd * width / 2
    verts = np.concatenate([np.concatenate([p0, p2, p1], 1), np.concatenate([p0, p3, p2], 1)]).astype(np.float32)
    sc = scenes.cornell_box(64, 64)
    return dict(sc, verts=verts, normals=scenes.face_normals(verts), mat_id=np.zeros(len(verts), np.int32))


def grazing_rays(sc, n, seed=9, cmin=1e-4, cmax=1e-1):
    """n rays aimed at random points of random triangles of `sc`, |cos(ray, triangle normal)| log-uniform in [cmin, cmax]
    (the reference rejects below 1e-4, Triangle.hpp:36), from 5..900 units away"""
    r = np.random.default_rng(seed)
    V = np.asarray(sc["verts"], np.float64).reshape(-1, 3, 3)
    k = r.integers(0, len(V), n)
    w = r.dirichlet([1, 1, 1], n)
    p = (V[k] * w[:, :, None]).sum(1)
    nrm = np.cross(V[k, 1] - V[k, 0], V[k, 2] - V[k, 0])
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    e = np.cross(nrm, r.normal(size=(n, 3)))
    e /= np.linalg.norm(e, axis=1, keepdims=True)
    c = np.exp(r.uniform(np.log(cmin), np.log(cmax), n)) * np.sign(r.uniform(-1, 1, n))
    d = e + c[:, None] * nrm
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    s = r.uniform(5, 900, n)
    return (p - d * s[:, None]).astype(np.float32), d.astype(np.float32)


def hit_conditioning(sc, O, D, tri, t):
    """float64 facts about reported hits (rows with tri >= 0): rel = (true crossing distance - reported fp32 t) / t (> 0: the fp32
    hit lies in FRONT of where the ray crosses the triangle's plane), kappa = |E1| |E2| |d| / |d . (E1 x E2)| =
    1 / (sin(angle at v0) |cos(ray, normal)|), the amplification of the triangle test's rounding errors,
    and inside = the true crossing point lies in the triangle (barycentrics >= 0 in float64)."""
    V = np.asarray(sc["verts"], np.float64).reshape(-1, 3, 3)
    o, d, tf = O.astype(np.float64), D.astype(np.float64), t.astype(np.float64)
    v0, e1, e2 = V[tri, 0], V[tri, 1] - V[tri, 0], V[tri, 2] - V[tri, 0]
    n = np.cross(e1, e2)
    det = (d * n).sum(1)
    tt = ((v0 - o) * n).sum(1) / det
    kappa = np.linalg.norm(e1, axis=1) * np.linalg.norm(e2, axis=1) * np.linalg.norm(d, axis=1) / np.abs(det)
    x = o + tt[:, None] * d - v0
    nn = (n * n).sum(1)
    b1 = (np.cross(x, e2) * n).sum(1) / nn
    b2 = (np.cross(e1, x) * n).sum(1) / nn
    inside = (b1 >= 0) & (b2 >= 0) & (1 - b1 - b2 >= 0)
    return (tt - tf) / np.maximum(np.abs(tf), 1e-30), kappa, inside
