"""TEST INFRASTRUCTURE ONLY -- ctypes front-end for the two CPU checkers declared in oracle/oracle_abi.h.

    Oracle("port")       -> oracle/libtutu_oracle.so   (our CPU restatement)
    Oracle("reference")  -> oracle/_ref/libtutu_ref.so (the reference's own code behind the same ABI)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIBS = {
    "port": os.path.join(HERE, "libtutu_oracle.so"),
    "reference": os.path.join(HERE, "_ref", "libtutu_ref.so"),
    # the same harness built -O3 for TIMING only (bench.py cpu_baseline); never used as a checker
    "reference_fast": os.path.join(HERE, "_ref", "libtutu_ref_fast.so"),
    # the same harness WITHOUT the RNG engine swap of ref_shim.h: the reference's own thread_local std::mt19937
    # (global.hpp:182-199) draws.  Only oracle/gen_frames.py uses it, for the statistical pin (SURVEY.md 8d parity (ii))
    "reference_native": os.path.join(HERE, "_ref", "libtutu_ref_native.so"),
}

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
assert MAT_DTYPE.itemsize == 56

LAMBERTIAN, PERFECT_REFLECTIVE, PERFECT_REFRACTIVE, MICROFACET_R, MICROFACET_T, UNLIT = range(6)


def make_material(type=LAMBERTIAN, diffuse=(0.9, 0.9, 0.9), emission=(0, 0, 0), alpha=1.0, eta=1.0, roughness=1.0,
                  metallic=0.0, specular=(1, 1, 1)):
    """Defaults = the reference's Material member initialisers (Material.hpp:21-30)."""
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


class _SceneDesc(C.Structure):
    _fields_ = [
        ("n_tris", C.c_int32),
        ("verts", C.c_void_p),
        ("normals", C.c_void_p),
        ("mat_id", C.c_void_p),
        ("n_mats", C.c_int32),
        ("mats", C.c_void_p),
        ("eta", C.c_float),
        ("bkg", C.c_float * 3),
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("eye", C.c_float * 3),
        ("viewdir", C.c_float * 3),
        ("updir", C.c_float * 3),
        ("hfov", C.c_int32),
        ("uvs", C.c_void_p),
        ("tex_ids", C.c_void_p),
        ("n_textures", C.c_int32 * 4),
        ("textures", C.c_void_p * 4),
        ("n_spheres", C.c_int32),
        ("spheres", C.c_void_p),
        ("sphere_mat_id", C.c_void_p),
        ("sphere_tex_ids", C.c_void_p),
        ("sphere_pos", C.c_void_p),
    ]


class _Texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgb", C.c_void_p)]


TEXTURE_LISTS = ("diffuse", "normal", "roughness", "metallic")


def pack_spheres(scene, desc):
    """scene["spheres"] (n,4) centre+radius, scene["sphere_mat_id"] (n,), optional scene["sphere_tex_ids"] (n,4) and
    scene["sphere_pos"] (n,) object-list positions -> fills the optional sphere fields of a scene descriptor"""
    if scene.get("spheres") is None or len(scene["spheres"]) == 0:
        return []
    sp = np.ascontiguousarray(scene["spheres"], dtype=np.float32).reshape(-1, 4)
    mid = np.ascontiguousarray(scene["sphere_mat_id"], dtype=np.int32)
    keep = [sp, mid]
    desc.n_spheres, desc.spheres, desc.sphere_mat_id = len(sp), sp.ctypes.data, mid.ctypes.data
    if scene.get("sphere_tex_ids") is not None:
        t = np.ascontiguousarray(scene["sphere_tex_ids"], dtype=np.int32).reshape(-1, 4)
        desc.sphere_tex_ids = t.ctypes.data
        keep.append(t)
    if scene.get("sphere_pos") is not None:
        q = np.ascontiguousarray(scene["sphere_pos"], dtype=np.int32)
        desc.sphere_pos = q.ctypes.data
        keep.append(q)
    return keep


def pack_textures(scene, desc, texture_struct):
    """scene["textures"] = {"diffuse": [HxWx3 float32, ...], "normal": [...], "roughness": [...], "metallic": [...]},
    scene["uvs"] (n,6), scene["tex_ids"] (n,4) -> fills the optional texture fields of a scene descriptor; returns
    the objects that must stay alive."""
    keep = []
    if scene.get("uvs") is None or scene.get("tex_ids") is None:
        return keep
    uvs = np.ascontiguousarray(scene["uvs"], dtype=np.float32).reshape(-1, 6)
    ids = np.ascontiguousarray(scene["tex_ids"], dtype=np.int32).reshape(-1, 4)
    desc.uvs, desc.tex_ids = uvs.ctypes.data, ids.ctypes.data
    keep += [uvs, ids]
    texs = scene.get("textures", {})
    for k, name in enumerate(TEXTURE_LISTS):
        lst = texs.get(name, [])
        arr = (texture_struct * max(1, len(lst)))()
        for i, img in enumerate(lst):
            a = np.ascontiguousarray(img, dtype=np.float32)
            arr[i].height, arr[i].width = a.shape[0], a.shape[1]
            arr[i].rgb = a.ctypes.data
            keep.append(a)
        desc.n_textures[k] = len(lst)
        desc.textures[k] = C.cast(arr, C.c_void_p).value
        keep.append(arr)
    return keep


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def available(kind):
    return os.path.exists(LIBS[kind])


class Oracle:
    def __init__(self, kind="port"):
        path = LIBS[kind]
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} is not built (run `make -C oracle`)")
        self.kind = kind
        self.lib = C.CDLL(path)
        self.lib.tor_kind.restype = C.c_char_p
        assert self.lib.tor_kind().decode() == kind.split("_")[0]

    # ---------------------------------------------------------------- RNG
    def philox(self, ctr4, key0, key1):
        ctr4 = np.ascontiguousarray(ctr4, dtype=np.uint32).reshape(-1, 4)
        out = np.empty_like(ctr4)
        self.lib.tor_philox4x32_10(C.c_int(len(ctr4)), _p(ctr4), C.c_uint32(key0), C.c_uint32(key1), _p(out))
        return out

    def rng_stream(self, pix, smp, key0, key1, n):
        xi = np.empty(n, np.float32)
        self.lib.tor_rng_stream(C.c_uint32(pix), C.c_uint32(smp), C.c_uint32(key0), C.c_uint32(key1), C.c_int(n), _p(xi))
        return xi

    # ---------------------------------------------------------------- pure functions
    def bbox_intersect(self, pmin, pmax, o, d):
        pmin, pmax, o, d = map(_f32, (pmin, pmax, o, d))
        n = len(o)
        hit = np.empty(n, np.uint8)
        self.lib.tor_bbox_intersect(C.c_int(n), _p(pmin), _p(pmax), _p(o), _p(d), _p(hit))
        return hit

    def tri_intersect(self, verts9, normals9, o, d):
        verts9, normals9, o, d = map(_f32, (verts9, normals9, o, d))
        n = len(o)
        hit = np.empty(n, np.uint8)
        t = np.empty(n, np.float32)
        pos = np.empty((n, 3), np.float32)
        Ns = np.empty((n, 3), np.float32)
        Ng = np.empty((n, 3), np.float32)
        self.lib.tor_tri_intersect(C.c_int(n), _p(verts9), _p(normals9), _p(o), _p(d), _p(hit), _p(t), _p(pos), _p(Ns), _p(Ng))
        return hit, t, pos, Ns, Ng

    def tri_area(self, verts9):
        verts9 = _f32(verts9).reshape(-1, 9)
        out = np.empty(len(verts9), np.float32)
        self.lib.tor_tri_area(C.c_int(len(verts9)), _p(verts9), _p(out))
        return out

    def _vec1(self, fn, *arrs, out_cols=3):
        arrs = [_f32(a) for a in arrs]
        n = len(arrs[0])
        out = np.empty((n, 3), np.float32) if out_cols == 3 else np.empty(n, np.float32)
        fn(C.c_int(n), *[_p(a) for a in arrs], _p(out))
        return out

    def normalized(self, v):
        return self._vec1(self.lib.tor_math_normalized, v)

    def fresnel(self, I, N, eta_i, eta_t):
        return self._vec1(self.lib.tor_math_fresnel, I, N, eta_i, eta_t, out_cols=1)

    def fresnel_schlick(self, cos_theta, F0):
        return self._vec1(self.lib.tor_math_fresnel_schlick, cos_theta, F0)

    def reflect(self, I, N):
        return self._vec1(self.lib.tor_math_reflect, I, N)

    def refract(self, I, N, eta_i, eta_t):
        return self._vec1(self.lib.tor_math_refract, I, N, eta_i, eta_t)

    def D(self, h, nrm, rough):
        return self._vec1(self.lib.tor_math_D, h, nrm, rough, out_cols=1)

    def G(self, wi, wo, nrm, rough, h):
        return self._vec1(self.lib.tor_math_G, wi, wo, nrm, rough, h, out_cols=1)

    def mis(self, a, b):
        return self._vec1(self.lib.tor_math_mis, a, b, out_cols=1)

    def local2world(self, N, d):
        return self._vec1(self.lib.tor_math_local2world, N, d)

    def texture_lookup(self, img, u, v):
        img = np.ascontiguousarray(img, dtype=np.float32)
        t = _Texture()
        t.height, t.width, t.rgb = img.shape[0], img.shape[1], img.ctypes.data
        u, v = _f32(u), _f32(v)
        out = np.empty((len(u), 3), np.float32)
        self.lib.tor_texture_lookup(C.byref(t), C.c_int(len(u)), _p(u), _p(v), _p(out))
        return out

    def postprocess(self, stage, img):
        """Postprocessor.hpp: 0 = performPostProcess (HDR_BLOOM), 1 = emissive, 2 = Gaussian blur, 3 = tone map"""
        img = np.ascontiguousarray(img, dtype=np.float32)
        out = np.empty_like(img)
        rc = self.lib.tor_postprocess(C.c_int(stage), C.c_int(img.shape[1]), C.c_int(img.shape[0]), _p(img), _p(out))
        assert rc == 0, rc
        return out

    def write_pixel(self, c):
        c = _f32(c).ravel()
        out = np.empty(len(c), np.int32)
        self.lib.tor_write_pixel(C.c_int(len(c)), _p(c), _p(out))
        return out

    # ---------------------------------------------------------------- material
    def mat_bxdf(self, mat, wi, wo, Ng, Ns, eta_scene=1.0, tir=None):
        wi, wo, Ng, Ns = map(_f32, (wi, wo, Ng, Ns))
        n = len(wi)
        m = np.ascontiguousarray(mat, dtype=MAT_DTYPE)
        out = np.empty((n, 3), np.float32)
        tirp = None
        if tir is not None:
            tir = np.ascontiguousarray(tir, dtype=np.uint8)
            tirp = _p(tir)
        self.lib.tor_mat_bxdf(C.c_int(n), _p(m), _p(wi), _p(wo), _p(Ng), _p(Ns), C.c_float(eta_scene), tirp, _p(out))
        return out

    def mat_pdf(self, mat, wi, wo, N, eta_i=1.0, eta_t=None):
        wi, wo, N = map(_f32, (wi, wo, N))
        n = len(wi)
        m = np.ascontiguousarray(mat, dtype=MAT_DTYPE)
        if eta_t is None:
            eta_t = float(m["eta"])
        out = np.empty(n, np.float32)
        self.lib.tor_mat_pdf(C.c_int(n), _p(m), _p(wi), _p(wo), _p(N), C.c_float(eta_i), C.c_float(eta_t), _p(out))
        return out

    def mat_sample(self, mat, wo, N, xi3, eta_i=1.0):
        wo, N, xi3 = map(_f32, (wo, N, xi3))
        n = len(wo)
        m = np.ascontiguousarray(mat, dtype=MAT_DTYPE)
        wi = np.zeros((n, 3), np.float32)
        ok = np.empty(n, np.uint8)
        sp = np.empty(n, np.uint8)
        nd = np.empty(n, np.int32)
        self.lib.tor_mat_sample(C.c_int(n), _p(m), _p(wo), _p(N), C.c_float(eta_i), _p(xi3), _p(wi), _p(ok), _p(sp), _p(nd))
        return wi, ok, sp, nd

    # ---------------------------------------------------------------- scene
    def scene(self, scene):
        return OracleScene(self, scene)

    def ref_load_obj(self, path, cap=200000):
        assert self.kind == "reference"
        v = np.zeros((cap, 9), np.float32)
        nr = np.zeros((cap, 9), np.float32)
        n = C.c_int32(0)
        rc = self.lib.tor_ref_load_obj(path.encode(), C.c_int(cap), C.byref(n), _p(v), _p(nr))
        if rc != 0:
            return None
        return v[: n.value].copy(), nr[: n.value].copy()


class OracleScene:
    """scene: dict with verts (n,9), normals (n,9), mat_id (n,), mats (MAT_DTYPE array), eta, bkg, width, height,
    eye, viewdir, updir, hfov."""

    def __init__(self, oracle, scene):
        self.o = oracle
        self.lib = oracle.lib
        self.verts = _f32(scene["verts"]).reshape(-1, 9)
        self.normals = _f32(scene["normals"]).reshape(-1, 9)
        self.mat_id = np.ascontiguousarray(scene["mat_id"], dtype=np.int32)
        self.mats = np.ascontiguousarray(scene["mats"], dtype=MAT_DTYPE)
        self.W = int(scene["width"])
        self.H = int(scene["height"])
        d = _SceneDesc()
        d.n_tris = len(self.verts)
        d.verts = self.verts.ctypes.data
        d.normals = self.normals.ctypes.data
        d.mat_id = self.mat_id.ctypes.data
        d.n_mats = len(self.mats)
        d.mats = self.mats.ctypes.data
        d.eta = float(scene.get("eta", 1.0))
        d.bkg = (C.c_float * 3)(*[float(x) for x in scene.get("bkg", (0, 0, 0))])
        d.width = self.W
        d.height = self.H
        d.eye = (C.c_float * 3)(*[float(x) for x in scene["eye"]])
        d.viewdir = (C.c_float * 3)(*[float(x) for x in scene["viewdir"]])
        d.updir = (C.c_float * 3)(*[float(x) for x in scene["updir"]])
        d.hfov = int(scene["hfov"])
        self._tex_keep = pack_textures(scene, d, _Texture)
        self._sphere_keep = pack_spheres(scene, d)
        self.h = C.c_void_p()
        rc = self.lib.tor_scene_create(C.byref(d), C.byref(self.h))
        if rc != 0:
            raise RuntimeError(f"tor_scene_create failed: {rc}")

    def close(self):
        if self.h:
            self.lib.tor_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bvh_dump(self):
        n_obj = len(self.verts) + (len(self._sphere_keep[0]) if self._sphere_keep else 0)
        cap = 2 * n_obj + 8
        b = np.zeros((cap, 6), np.float32)
        leaf = np.zeros(cap, np.int32)
        n = C.c_int32(0)
        self.lib.tor_scene_bvh_dump(self.h, C.c_int(cap), C.byref(n), _p(b), _p(leaf))
        return b[: n.value].copy(), leaf[: n.value].copy()

    def closest(self, o, d):
        o, d = _f32(o), _f32(d)
        n = len(o)
        hit = np.empty(n, np.uint8)
        t = np.empty(n, np.float32)
        tri = np.empty(n, np.int32)
        pos = np.empty((n, 3), np.float32)
        Ns = np.empty((n, 3), np.float32)
        Ng = np.empty((n, 3), np.float32)
        self.lib.tor_scene_closest(self.h, C.c_int(n), _p(o), _p(d), _p(hit), _p(t), _p(tri), _p(pos), _p(Ns), _p(Ng))
        return hit, t, tri, pos, Ns, Ng

    def any_hit(self, orig, target):
        orig, target = _f32(orig), _f32(target)
        n = len(orig)
        b = np.empty(n, np.uint8)
        self.lib.tor_scene_any(self.h, C.c_int(n), _p(orig), _p(target), _p(b))
        return b

    def lights(self):
        cap = len(self.verts)
        tri = np.zeros(cap, np.int32)
        n = C.c_int32(0)
        self.lib.tor_scene_lights(self.h, C.c_int(cap), C.byref(n), _p(tri))
        return tri[: n.value].copy()

    def sample_light(self, xi3):
        xi3 = _f32(xi3)
        n = len(xi3)
        tri = np.empty(n, np.int32)
        pos = np.empty((n, 3), np.float32)
        nrm = np.empty((n, 3), np.float32)
        pdf = np.empty(n, np.float32)
        self.lib.tor_scene_sample_light(self.h, C.c_int(n), _p(xi3), _p(tri), _p(pos), _p(nrm), _p(pdf))
        return tri, pos, nrm, pdf

    def light_pdf(self, tri):
        tri = np.ascontiguousarray(tri, dtype=np.int32)
        out = np.empty(len(tri), np.float32)
        self.lib.tor_scene_light_pdf(self.h, C.c_int(len(tri)), _p(tri), _p(out))
        return out

    def camera(self):
        out = np.empty(18, np.float32)
        self.lib.tor_camera(self.h, _p(out))
        return out.reshape(6, 3)

    def camera_raster(self):
        out = np.empty(22, np.float32)
        self.lib.tor_camera_raster(self.h, _p(out))
        return out

    def raydir(self, px, py):
        px = np.ascontiguousarray(px, dtype=np.int32)
        py = np.ascontiguousarray(py, dtype=np.int32)
        d = np.empty((len(px), 3), np.float32)
        self.lib.tor_camera_raydir(self.h, C.c_int(len(px)), _p(px), _p(py), _p(d))
        return d

    def trace_samples(self, pix, smp, key0, key1, stats=False):
        pix = np.ascontiguousarray(pix, dtype=np.uint32)
        smp = np.ascontiguousarray(smp, dtype=np.uint32)
        n = len(pix)
        L = np.empty((n, 3), np.float32)
        nd = np.empty(n, np.int32)
        nc = np.empty(n, np.int32)
        self.lib.tor_trace_samples(self.h, C.c_int(n), _p(pix), _p(smp), C.c_uint32(key0), C.c_uint32(key1), _p(L), _p(nd), _p(nc))
        return (L, nd, nc) if stats else L

    def path_stats(self, pix, smp, key0, key1, ordered=True):
        """port only: work counters of the ordered, t-pruned traversal (the HIP kernels' algorithm) over the given
        samples -- SURVEY.md 8(d)'s segs / rays / N / T."""
        assert self.o.kind == "port"
        pix = np.ascontiguousarray(pix, dtype=np.uint32)
        smp = np.ascontiguousarray(smp, dtype=np.uint32)
        out = np.zeros(8, np.int64)
        self.lib.tor_port_set_ordered(self.h, C.c_int(1 if ordered else 0))
        self.lib.tor_port_path_stats(self.h, C.c_int(len(pix)), _p(pix), _p(smp), C.c_uint32(key0), C.c_uint32(key1), _p(out))
        self.lib.tor_port_set_ordered(self.h, C.c_int(0))
        n, segs, closest, shadow, nodes, tris, nodes_sh, tris_sh = [int(x) for x in out]
        return {"samples": n, "segs": segs / n, "closest": closest / n, "shadow": shadow / n,
                "N_closest": (nodes - nodes_sh) / max(closest, 1), "T_closest": (tris - tris_sh) / max(closest, 1),
                "N_shadow": nodes_sh / max(shadow, 1), "T_shadow": tris_sh / max(shadow, 1),
                "N_all": nodes / max(closest + shadow, 1), "T_all": tris / max(closest + shadow, 1)}

    def render_integrator(self, itype, spp, key0, key1):
        """LightTracing (1) / NaivePT (2) / BDPT (3): a whole frame in the reference's own loop order, single-threaded, one
        sequential random stream"""
        rgb = np.zeros((self.H, self.W, 3), np.float32)
        rc = self.lib.tor_render_integrator(self.h, C.c_int(itype), C.c_int(spp), C.c_uint32(key0), C.c_uint32(key1), _p(rgb))
        if rc != 0:
            raise RuntimeError(f"tor_render_integrator failed: {rc}")
        return rgb

    def render_integrator_units(self, itype, spp, key0, key1):
        """port only: the same frame from per-(pixel, sample) streams -- what the HIP integrators render"""
        rgb = np.zeros((self.H, self.W, 3), np.float32)
        rc = self.lib.tor_render_integrator_units(self.h, C.c_int(itype), C.c_int(spp), C.c_uint32(key0), C.c_uint32(key1), C.c_int(1), _p(rgb))
        if rc != 0:
            raise RuntimeError(f"tor_render_integrator_units failed: {rc}")
        return rgb

    def integrator_samples(self, itype, spp, pix, smp, key0, key1, max_ev=8):
        """port only: per-unit outputs -- own (n,3), alive (n,), n_ev (n,), ev_op / ev_index (n,max_ev), ev_rgb (n,max_ev,3)"""
        pix = np.ascontiguousarray(pix, dtype=np.uint32)
        smp = np.ascontiguousarray(smp, dtype=np.uint32)
        n = len(pix)
        own = np.zeros((n, 3), np.float32)
        alive = np.zeros(n, np.uint8)
        n_ev = np.zeros(n, np.int32)
        op = np.full((n, max_ev), -1, np.int32)
        idx = np.full((n, max_ev), -1, np.int32)
        rgb = np.zeros((n, max_ev, 3), np.float32)
        rc = self.lib.tor_integrator_samples(self.h, C.c_int(itype), C.c_int(spp), C.c_int(n), _p(pix), _p(smp), C.c_uint32(key0),
                                             C.c_uint32(key1), _p(own), _p(alive), C.c_int(max_ev), _p(n_ev), _p(op), _p(idx), _p(rgb))
        if rc != 0:
            raise RuntimeError(f"tor_integrator_samples failed: {rc}")
        return own, alive, n_ev, op, idx, rgb

    def render(self, spp, key0, key1, rect=None, nthreads=None):
        if rect is None:
            rect = (0, 0, self.W, self.H)
        if nthreads is None:
            nthreads = os.cpu_count() or 1
        rgb = np.zeros((self.H, self.W, 3), np.float32)
        rc = self.lib.tor_render(self.h, C.c_int(spp), C.c_uint32(key0), C.c_uint32(key1), C.c_int(rect[0]), C.c_int(rect[1]),
                                 C.c_int(rect[2]), C.c_int(rect[3]), C.c_int(nthreads), _p(rgb))
        if rc != 0:
            raise RuntimeError(f"tor_render failed: {rc}")
        return rgb
