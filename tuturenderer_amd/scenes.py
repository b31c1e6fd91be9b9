"""Scene programs: the Python-side equivalent of the reference's `src/main_*.cpp`.

The reference does not describe geometry in `config.txt`; every scene is hard-coded in a `main` that loads OBJ
files and attaches `Material`s (src/main_cornellBox.cpp:23-71, src/main_veach_bdpt.cpp:24-86).  This module does
the same job for the benchmark configs of BASELINE.json and returns plain arrays that go straight into the C ABI
(`include/tutu_hip.h`, `TutuSceneDesc`).

A scene is a dict:
    verts   (n,9) float32  v0 v1 v2 per triangle, in the reference's load order
    normals (n,9) float32  n0 n1 n2 per triangle (un-normalised when generated, as OBJ_Loader.h:818-836 leaves them)
    mat_id  (n,)  int32
    mats    (m,)  MAT_DTYPE (56 B, same field order as the reference's Material, Material.hpp:19-30)
    eta, bkg, width, height, eye, viewdir, updir, hfov   -- the `config.txt` keywords
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = os.path.join(HERE, "scenes")

MAT_DTYPE = np.dtype(
    [
        ("diffuse", "<f4", 3),
        ("specular", "<f4", 3),
        ("emission", "<f4", 3),
        ("type", "<i4"),
        ("alpha", "<f4"),
        ("eta", "<f4"),
        ("roughness", "<f4"),
        ("metallic", "<f4"),
    ]
)

# MaterialType, Material.hpp:9-16
LAMBERTIAN, PERFECT_REFLECTIVE, PERFECT_REFRACTIVE, MICROFACET_R, MICROFACET_T, UNLIT = range(6)


def material(type=LAMBERTIAN, diffuse=(0.9, 0.9, 0.9), emission=(0.0, 0.0, 0.0), alpha=1.0, eta=1.0, roughness=1.0,
             metallic=0.0, specular=(1.0, 1.0, 1.0)):
    """Defaults are the reference's member initialisers (Material.hpp:21-30)."""
    m = np.zeros((), dtype=MAT_DTYPE)
    m["diffuse"] = diffuse
    m["specular"] = specular
    m["emission"] = emission
    m["type"] = type
    m["alpha"] = alpha
    m["eta"] = eta
    m["roughness"] = roughness
    m["metallic"] = metallic
    return m


def face_normals(verts):
    """Per-face normal the reference generates for OBJ faces without `vn` (OBJ_Loader.h:818-836):
    (v1-v0) x (v2-v1), NOT normalised, copied to the three vertices."""
    v = np.asarray(verts, np.float32).reshape(-1, 3, 3)
    a = v[:, 1] - v[:, 0]
    b = v[:, 2] - v[:, 1]
    n = np.empty_like(a)
    n[:, 0] = a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1]
    n[:, 1] = a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2]
    n[:, 2] = a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]
    return np.repeat(n[:, None, :], 3, axis=1).reshape(-1, 9).astype(np.float32)


def _quads(pts, pattern="fan"):
    """Quads (groups of four points) -> triangles in the face order the Cornell OBJ files use:
    'fan'  : (1 2 3), (1 3 4)      light/left/right/boxes
    'strip': (1 2 3), (3 4 1)      floor.obj"""
    p = np.asarray(pts, np.float32).reshape(-1, 4, 3)
    if pattern == "fan":
        idx = [(0, 1, 2), (0, 2, 3)]
    else:
        idx = [(0, 1, 2), (2, 3, 0)]
    tris = []
    for q in p:
        for i in idx:
            tris.append(np.concatenate([q[i[0]], q[i[1]], q[i[2]]]))
    return np.asarray(tris, np.float32)


# The Cornell box measurements (the data of model/cornellBox/*.obj, loaded in the order of
# src/main_cornellBox.cpp:27-71: floor(+ceiling+back wall), light, right(green), left(red), tall box, short box).
_CB_FLOOR = [
    (552.8, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, 559.2), (549.6, 0.0, 559.2),            # floor
    (556.0, 548.8, 0.0), (556.0, 548.8, 559.2), (0.0, 548.8, 559.2), (0.0, 548.8, 0.0),      # ceiling
    (549.6, 0.0, 559.2), (0.0, 0.0, 559.2), (0.0, 548.8, 559.2), (556.0, 548.8, 559.2),      # back wall
]
_CB_LIGHT = [(343.0, 548.7, 227.0), (343.0, 548.7, 332.0), (213.0, 548.7, 332.0), (213.0, 548.7, 227.0)]
_CB_RIGHT = [(0.0, 0.0, 559.2), (0.0, 0.0, 0.0), (0.0, 548.8, 0.0), (0.0, 548.8, 559.2)]
_CB_LEFT = [(552.8, 0.0, 0.0), (549.6, 0.0, 559.2), (556.0, 548.8, 559.2), (556.0, 548.8, 0.0)]
_CB_TALL = [
    (423.0, 330.0, 247.0), (265.0, 330.0, 296.0), (314.0, 330.0, 456.0), (472.0, 330.0, 406.0),
    (423.0, 0.0, 247.0), (423.0, 330.0, 247.0), (472.0, 330.0, 406.0), (472.0, 0.0, 406.0),
    (472.0, 0.0, 406.0), (472.0, 330.0, 406.0), (314.0, 330.0, 456.0), (314.0, 0.0, 456.0),
    (314.0, 0.0, 456.0), (314.0, 330.0, 456.0), (265.0, 330.0, 296.0), (265.0, 0.0, 296.0),
    (265.0, 0.0, 296.0), (265.0, 330.0, 296.0), (423.0, 330.0, 247.0), (423.0, 0.0, 247.0),
]
_CB_SHORT = [
    (130.0, 165.0, 65.0), (82.0, 165.0, 225.0), (240.0, 165.0, 272.0), (290.0, 165.0, 114.0),
    (290.0, 0.0, 114.0), (290.0, 165.0, 114.0), (240.0, 165.0, 272.0), (240.0, 0.0, 272.0),
    (130.0, 0.0, 65.0), (130.0, 165.0, 65.0), (290.0, 165.0, 114.0), (290.0, 0.0, 114.0),
    (82.0, 0.0, 225.0), (82.0, 165.0, 225.0), (130.0, 165.0, 65.0), (130.0, 0.0, 65.0),
    (240.0, 0.0, 272.0), (240.0, 165.0, 272.0), (82.0, 165.0, 225.0), (82.0, 0.0, 225.0),
]

CB_WHITE = (0.725, 0.71, 0.68)
CB_GREEN = (0.14, 0.45, 0.091)
CB_RED = (0.63, 0.065, 0.05)
CB_EMISSION = (47.8348007, 38.5663986, 31.0807991)


def cornell_parts():
    """[(name, verts(n,9))] in load order."""
    return [
        ("floor", _quads(_CB_FLOOR, "strip")),
        ("light", _quads(_CB_LIGHT)),
        ("right", _quads(_CB_RIGHT)),
        ("left", _quads(_CB_LEFT)),
        ("tallbox", _quads(_CB_TALL)),
        ("shortbox", _quads(_CB_SHORT)),
    ]


def _camera(scene, width, height, eye, viewdir, updir=(0, 1, 0), hfov=40, bkg=(0, 0, 0), eta=1.0):
    scene.update(dict(width=int(width), height=int(height), eye=tuple(eye), viewdir=tuple(viewdir), updir=tuple(updir),
                      hfov=int(hfov), bkg=tuple(bkg), eta=float(eta)))
    return scene


def _assemble(parts, mats):
    verts = np.concatenate([p[0] for p in parts]).astype(np.float32)
    normals = np.concatenate([p[1] for p in parts]).astype(np.float32)
    mat_id = np.concatenate([np.full(len(p[0]), p[2], np.int32) for p in parts])
    return dict(verts=verts, normals=normals, mat_id=mat_id, mats=np.array(mats, dtype=MAT_DTYPE))


def cornell_box(width=800, height=800, tall=None, short=None):
    """BASELINE configs 1/2: src/main_cornellBox.cpp + configs/config_cornellBox.txt (imsize overridden).
    `tall` / `short` optionally replace the two boxes' material (used by the material-coverage tests)."""
    mats = [
        material(LAMBERTIAN, CB_WHITE),                          # 0 floor / ceiling / back wall / boxes
        material(LAMBERTIAN, CB_WHITE, emission=CB_EMISSION),    # 1 light (mType stays the default LAMBERTIAN)
        material(LAMBERTIAN, CB_GREEN),                          # 2 right.obj
        material(LAMBERTIAN, CB_RED),                            # 3 left.obj
    ]
    part_mat = {"floor": 0, "light": 1, "right": 2, "left": 3, "tallbox": 0, "shortbox": 0}
    if tall is not None:
        mats.append(tall)
        part_mat["tallbox"] = len(mats) - 1
    if short is not None:
        mats.append(short)
        part_mat["shortbox"] = len(mats) - 1
    parts = [(v, face_normals(v), part_mat[name]) for name, v in cornell_parts()]
    s = _assemble(parts, mats)
    return _camera(s, width, height, eye=(278, 273, -800), viewdir=(0, 0, 1))


def veach_room(width=800, height=600, small_light=False):
    """BASELINE config 5: src/main_veach_bdpt.cpp:24-86 rendered with the PathTracing integrator, camera from
    configs/config_veach_bdpt.txt.  Geometry comes from scenes/veach_room.npz, a dump of the reference's own
    OBJ loading of model/veach_bdpt/*.obj (written by oracle/gen_golden.py).  On a case-sensitive file system
    the reference silently skips the small light (it asks for `veach_slight.obj`, the file is
    `veach_sLight.obj`); small_light=False reproduces that, True loads it."""
    z = np.load(os.path.join(DATA, "veach_room.npz"))
    room = material(LAMBERTIAN, CB_WHITE)
    e = np.float32(500.0) * np.float32(0.5)
    llight = material(LAMBERTIAN, CB_WHITE, emission=(e, e, e))
    se = np.array([6999.999881, 5450.000167, 3630.000055], np.float32) * np.float32(0.5)
    slight = material(LAMBERTIAN, CB_WHITE, emission=tuple(se))
    brown = (0.32962962985, 0.257976263762, 0.150291711092)
    table = material(LAMBERTIAN, brown)
    glass = material(PERFECT_REFRACTIVE, eta=1.5)
    lamp = material(MICROFACET_R, brown, roughness=0.2775146484375, metallic=0.5)
    mats = [room, llight, slight, table, glass, lamp]
    order = [("room", 0), ("Llight", 1)] + ([("sLight", 2)] if small_light else []) + \
            [("table", 3), ("glass", 4), ("tallLamp", 5), ("wallLamp", 0)]
    parts = [(z[name + "_verts"], z[name + "_normals"], mi) for name, mi in order]
    s = _assemble(parts, mats)
    return _camera(s, width, height, eye=(-0.5, 0, 7.6), viewdir=(-0.005, 0, -1))


# ------------------------------------------------------------------------------------------------------------
# Synthetic stand-ins for the two BASELINE configs whose assets are not in the reference repository
# (SURVEY.md 8(d)): C3 "bunny in the box" and C4 "broom".
def _philox_uniform(n, seed):
    rng = np.random.Generator(np.random.Philox(key=seed))
    return rng.random(n, dtype=np.float64)


def icosphere(subdiv):
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
                  (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)], np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2),
                  (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11),
                  (6, 2, 10), (8, 6, 7), (9, 8, 1)], np.int64)
    for _ in range(subdiv):
        edges = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
        key = np.sort(edges, axis=1)
        uniq, inv = np.unique(key, axis=0, return_inverse=True)
        mid = v[uniq[:, 0]] + v[uniq[:, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        base = len(v)
        v = np.concatenate([v, mid])
        nf = len(f)
        m01 = base + inv[:nf]
        m12 = base + inv[nf:2 * nf]
        m20 = base + inv[2 * nf:]
        f = np.concatenate([np.stack([f[:, 0], m01, m20], 1), np.stack([f[:, 1], m12, m01], 1),
                            np.stack([f[:, 2], m20, m12], 1), np.stack([m01, m12, m20], 1)])
    return v, f


def bumpy_blob(subdiv=6, radius=90.0, centre=(185.0, 90.0, 169.0), seed=70001):
    """Closed smooth-normal mesh: icosphere displaced radially by 1 + 0.25 * sum_k 2^-k sin(f_k . p + phi_k)."""
    v, f = icosphere(subdiv)
    u = _philox_uniform(16, seed)
    disp = np.ones(len(v))
    for k in range(1, 5):
        fk = (u[4 * (k - 1):4 * (k - 1) + 3] * 2 - 1) * (2.0 + 3.0 * k)
        phik = u[4 * (k - 1) + 3] * 2 * np.pi
        disp += 0.25 * (2.0 ** -k) * np.sin(v @ fk + phik)
    p = v * disp[:, None] * radius + np.asarray(centre)
    # smooth vertex normals = normalised sum of incident face normals
    fn = np.cross(p[f[:, 1]] - p[f[:, 0]], p[f[:, 2]] - p[f[:, 0]])
    vn = np.zeros_like(p)
    for c in range(3):
        np.add.at(vn, f[:, c], fn)
    vn /= np.linalg.norm(vn, axis=1, keepdims=True)
    verts = p[f].reshape(-1, 9).astype(np.float32)
    normals = vn[f].reshape(-1, 9).astype(np.float32)
    return verts, normals


def bunny_box(width=1024, height=1024, subdiv=6):
    """BASELINE config 3 stand-in (no bunny model exists in the reference): Cornell box whose short box is replaced
    by an 81,920-triangle displaced icosphere of rough glass (MICROFACET_T, eta 1.5, roughness 0.2)."""
    mats = [material(LAMBERTIAN, CB_WHITE), material(LAMBERTIAN, CB_WHITE, emission=CB_EMISSION),
            material(LAMBERTIAN, CB_GREEN), material(LAMBERTIAN, CB_RED),
            material(MICROFACET_T, CB_WHITE, eta=1.5, roughness=0.2, alpha=1.0)]
    part_mat = {"floor": 0, "light": 1, "right": 2, "left": 3, "tallbox": 0}
    parts = [(v, face_normals(v), part_mat[name]) for name, v in cornell_parts() if name != "shortbox"]
    bv, bn = bumpy_blob(subdiv)
    parts.append((bv, bn, 4))
    s = _assemble(parts, mats)
    return _camera(s, width, height, eye=(278, 273, -800), viewdir=(0, 0, 1))


def broom_room(width=1600, height=900, n_bristles=4000, seed=70002):
    """BASELINE config 4 stand-in (the broom scene is a missing blob): a 16:9 Cornell-shaped room with
    `n_bristles` thin six-sided Lambertian prisms (12 triangles each), a MICROFACET_R floor and one quad light."""
    sx = 16.0 / 9.0
    W, Hh, D = 556.0 * sx, 548.8, 559.2

    def quad(a, b, c, d):
        return _quads([a, b, c, d])

    floor = quad((W, 0, 0), (0, 0, 0), (0, 0, D), (W, 0, D))
    ceil = quad((W, Hh, 0), (W, Hh, D), (0, Hh, D), (0, Hh, 0))
    back = quad((W, 0, D), (0, 0, D), (0, Hh, D), (W, Hh, D))
    right = quad((0, 0, D), (0, 0, 0), (0, Hh, 0), (0, Hh, D))
    left = quad((W, 0, 0), (W, 0, D), (W, Hh, D), (W, Hh, 0))
    lx0, lx1 = W / 2 - 100, W / 2 + 100
    light = quad((lx1, Hh - 0.1, 227), (lx1, Hh - 0.1, 332), (lx0, Hh - 0.1, 332), (lx0, Hh - 0.1, 227))
    u = _philox_uniform(n_bristles * 6, seed).reshape(n_bristles, 6)
    tris = []
    ang = np.arange(6) * (np.pi / 3)
    for i in range(n_bristles):
        bx = 40 + u[i, 0] * (W - 80)
        bz = 120 + u[i, 1] * (D - 200)
        h = 120 + u[i, 2] * 220
        r = 1.0 + u[i, 3] * 1.5
        tilt = u[i, 4] * np.deg2rad(15.0)
        az = u[i, 5] * 2 * np.pi
        axis = np.array([np.sin(tilt) * np.cos(az), np.cos(tilt), np.sin(tilt) * np.sin(az)])
        b0 = np.array([bx, 0.01, bz])
        t0 = b0 + axis * h
        ring = np.stack([np.cos(ang), np.zeros(6), np.sin(ang)], 1) * r
        for k in range(6):
            a, b = ring[k], ring[(k + 1) % 6]
            p0, p1, p2, p3 = b0 + a, b0 + b, t0 + b, t0 + a
            tris.append(np.concatenate([p0, p2, p1]))
            tris.append(np.concatenate([p0, p3, p2]))
    bristles = np.asarray(tris, np.float32)
    mats = [material(LAMBERTIAN, CB_WHITE), material(LAMBERTIAN, CB_WHITE, emission=CB_EMISSION),
            material(LAMBERTIAN, CB_GREEN), material(LAMBERTIAN, CB_RED),
            material(MICROFACET_R, (0.6, 0.6, 0.6), roughness=0.3, metallic=0.5),
            material(LAMBERTIAN, (0.55, 0.42, 0.2))]
    parts = [(floor, 4), (ceil, 0), (back, 0), (light, 1), (right, 2), (left, 3), (bristles, 5)]
    parts = [(v, face_normals(v), m) for v, m in parts]
    s = _assemble(parts, mats)
    return _camera(s, width, height, eye=(W / 2, 273, -800 * sx * 0.62), viewdir=(0, 0, 1), hfov=50)


# ---- textures on triangles ------------------------------------------------------------------------
# A scene may carry, next to verts/normals/mat_id:
#   uvs      (n_tris, 6)  uv0 uv1 uv2 per triangle (the `vt` of the OBJ face corners)
#   tex_ids  (n_tris, 4)  index into the diffuse / normal / roughness / metallic map lists, -1 = none
#   textures {"diffuse": [HxWx3 f32, ...], "normal": [...], "roughness": [...], "metallic": [...]}
# Normal-map texels are stored already mapped to [-1,1] (the `bump` keyword does c*2-1 at load time,
# PPMGenerator.hpp:713-722); all other maps hold c/255 like loadTexture (PPMGenerator.hpp:1027-1084).
TEXTURE_LISTS = ("diffuse", "normal", "roughness", "metallic")


def _planar_uvs(verts, scale, offset=(0.0, 0.0)):
    """uv per corner from the two coordinates in which the triangle's plane is largest; deliberately reaches
    negative and >1 values so the wrap rule of Texture::getRGBat (Texture.hpp:18-39) is exercised."""
    v = np.asarray(verts, np.float32).reshape(-1, 3, 3)
    n = np.abs(np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]))
    drop = np.argmax(n, axis=1)
    keep = np.array([[1, 2], [0, 2], [0, 1]])[drop]
    uv = np.take_along_axis(v, keep[:, None, :].repeat(3, axis=1), axis=2)
    uv = uv * np.float32(scale) + np.asarray(offset, np.float32)
    return uv.reshape(-1, 6).astype(np.float32)


def procedural_maps(seed=70003):
    """Four small seeded maps (non-square on purpose): checker+noise albedo, a bumpy tangent-space normal map,
    a roughness map and a metallic map."""
    rng = np.random.Generator(np.random.Philox(key=seed))
    h, w = 48, 64
    yy, xx = np.mgrid[0:h, 0:w]
    check = (((xx // 8) + (yy // 8)) & 1).astype(np.float32)
    albedo = np.stack([0.25 + 0.6 * check, 0.3 + 0.4 * (1 - check), 0.2 + 0.5 * (xx / w)], -1).astype(np.float32)
    albedo = np.round((albedo + 0.1 * rng.random((h, w, 3), dtype=np.float32)).clip(0, 1) * 255) / np.float32(255)
    hn, wn = 40, 56
    nx = rng.random((hn, wn), dtype=np.float32) - 0.5
    ny = rng.random((hn, wn), dtype=np.float32) - 0.5
    c = np.round(np.stack([0.5 + 0.35 * nx, 0.5 + 0.35 * ny, np.full((hn, wn), 0.92, np.float32)], -1) * 255) / np.float32(255)
    normal = (c * np.float32(2) - np.float32(1)).astype(np.float32)
    rough = np.round((0.08 + 0.8 * rng.random((32, 24, 1), dtype=np.float32)) * 255) / np.float32(255)
    metal = np.round(rng.random((16, 20, 1), dtype=np.float32) * 255) / np.float32(255)
    return {
        "diffuse": [albedo.astype(np.float32), np.ascontiguousarray(albedo[::-1, :, ::-1]).astype(np.float32)],
        "normal": [normal],
        "roughness": [np.repeat(rough, 3, axis=2).astype(np.float32)],
        "metallic": [np.repeat(metal, 3, axis=2).astype(np.float32)],
    }


def cornell_textured(width=800, height=800):
    """The Cornell box with maps on it: albedo + normal map on the floor/ceiling/back wall, a second albedo on the
    short box, and a GGX-R tall box driven by roughness + metallic + normal maps.  Left/right walls and the light stay
    untextured, so both kinds of triangle are in one BVH."""
    mats = [
        material(LAMBERTIAN, CB_WHITE),
        material(LAMBERTIAN, CB_WHITE, emission=CB_EMISSION),
        material(LAMBERTIAN, CB_GREEN),
        material(LAMBERTIAN, CB_RED),
        material(MICROFACET_R, CB_WHITE, roughness=0.4, metallic=0.3),
    ]
    part_mat = {"floor": 0, "light": 1, "right": 2, "left": 3, "tallbox": 4, "shortbox": 0}
    part_tex = {"floor": (0, 0, -1, -1), "shortbox": (1, -1, -1, -1), "tallbox": (-1, 0, 0, 0)}
    parts, uvs, ids = [], [], []
    for name, v in cornell_parts():
        parts.append((v, face_normals(v), part_mat[name]))
        t = part_tex.get(name)
        if t is None:
            uvs.append(np.full((len(v), 6), -1.0, np.float32))  # Vector2f() default, Vector.hpp:54-57
            ids.append(np.full((len(v), 4), -1, np.int32))
        else:
            uvs.append(_planar_uvs(v, 1.0 / 170.0, (-0.8, 0.3)))
            ids.append(np.tile(np.array(t, np.int32), (len(v), 1)))
    s = _assemble(parts, mats)
    s["uvs"] = np.concatenate(uvs).astype(np.float32)
    s["tex_ids"] = np.concatenate(ids).astype(np.int32)
    s["textures"] = procedural_maps()
    return _camera(s, width, height, eye=(278, 273, -800), viewdir=(0, 0, 1))


# ---- spheres ----------------------------------------------------------------------------------------
# A scene may carry spheres next to its triangles:
#   spheres        (n, 4)  centre xyz, radius           sphere_mat_id (n,)
#   sphere_tex_ids (n, 4)  optional map indices         sphere_pos    (n,) optional position of each sphere in the
#                                                                      combined object list (default: after the triangles)
def cornell_spheres(width=800, height=800):
    """The Cornell room (no boxes) with five spheres: a mirror ball, a glass ball, a textured Lambertian ball (albedo +
    normal map), a rough-glass ball and a small emissive ball next to the ceiling light -- so closest-hit, any-hit,
    sphere light sampling, getLightPdf on a sphere and the sphere case of changeNormalDir are all on the path.
    The spheres sit at interleaved positions of the object list."""
    mats = [
        material(LAMBERTIAN, CB_WHITE),
        material(LAMBERTIAN, CB_WHITE, emission=CB_EMISSION),
        material(LAMBERTIAN, CB_GREEN),
        material(LAMBERTIAN, CB_RED),
        material(PERFECT_REFLECTIVE, CB_WHITE),
        material(PERFECT_REFRACTIVE, CB_WHITE, eta=1.5),
        material(MICROFACET_T, CB_WHITE, eta=1.5, roughness=0.25),
        material(LAMBERTIAN, CB_WHITE, emission=(30.0, 24.0, 12.0)),
    ]
    part_mat = {"floor": 0, "light": 1, "right": 2, "left": 3}
    parts = [(v, face_normals(v), part_mat[name]) for name, v in cornell_parts() if name in part_mat]
    s = _assemble(parts, mats)
    n_tris = len(s["verts"])
    s["uvs"] = np.full((n_tris, 6), -1.0, np.float32)
    s["tex_ids"] = np.full((n_tris, 4), -1, np.int32)
    s["textures"] = procedural_maps()
    s["spheres"] = np.array([
        [400.0, 90.0, 350.0, 90.0],    # mirror
        [150.0, 70.0, 200.0, 70.0],    # glass
        [290.0, 60.0, 120.0, 60.0],    # textured Lambertian
        [420.0, 45.0, 130.0, 45.0],    # rough glass
        [120.0, 420.0, 400.0, 25.0],   # emissive
    ], np.float32)
    s["sphere_mat_id"] = np.array([4, 5, 0, 6, 7], np.int32)
    s["sphere_tex_ids"] = np.array([[-1, -1, -1, -1], [-1, -1, -1, -1], [0, 0, -1, -1], [-1, -1, -1, -1], [-1, -1, -1, -1]], np.int32)
    s["sphere_pos"] = np.array([0, 5, 11, n_tris + 3, n_tris + 4], np.int32)
    return _camera(s, width, height, eye=(278, 273, -800), viewdir=(0, 0, 1))


SCENES = {
    "cornell": cornell_box,
    "veach": veach_room,
    "bunny": bunny_box,
    "broom": broom_room,
    "cornell_textured": cornell_textured,
    "cornell_spheres": cornell_spheres,
}
