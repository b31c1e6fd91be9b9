"""tuturenderer_amd -- MI355X-native path-tracing integrator, drop-in for TutuRenderer's PathTracing hot path.

Python front-end over the C ABI of include/tutu_hip.h (tuturenderer_amd/libtutu_hip.so, hand-written HIP for
gfx950).  There is no CPU fallback: if the library is not built, or no GPU is present, the calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import scenes  # noqa: F401
from .scenes import MAT_DTYPE

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TUTU_HIP_LIB", os.path.join(HERE, "libtutu_hip.so"))  # override: A/B runs of two builds on one box

# every symbol include/tutu_hip.h declares
ABI_SYMBOLS = [
    "tutu_camera_frame", "tutu_bvh_build_preorder", "tutu_host_wide8", "tutu_hip_error_string", "tutu_hip_last_error", "tutu_hip_version",
    "tutu_hip_device_count", "tutu_hip_create", "tutu_hip_destroy", "tutu_hip_render", "tutu_hip_render_device", "tutu_hip_render_multi",
    "tutu_hip_trace_closest", "tutu_hip_trace_any", "tutu_hip_trace_samples", "tutu_hip_eval_bxdf", "tutu_hip_eval_pdf",
    "tutu_hip_eval_sample", "tutu_hip_eval_sample_light", "tutu_hip_eval_fn", "tutu_hip_scene_info", "tutu_hip_eval_texture", "tutu_hip_set_option", "tutu_hip_get_option", "tutu_hip_postprocess", "tutu_hip_quantise",
    "tutu_camera_raster", "tutu_hip_render_integrator", "tutu_hip_integrator_samples", "tutu_hip_work_ready", "tutu_hip_render_multi_device",
]


class TutuError(RuntimeError):
    pass


class SceneDesc(C.Structure):
    _fields_ = [("n_tris", C.c_uint32), ("verts", C.c_void_p), ("normals", C.c_void_p), ("mat_id", C.c_void_p),
                ("n_mats", C.c_uint32), ("mats", C.c_void_p), ("eta", C.c_float), ("bkg", C.c_float * 3),
                ("textures", C.c_void_p), ("spheres", C.c_void_p)]


class SphereSet(C.Structure):
    _fields_ = [("n_spheres", C.c_uint32), ("spheres", C.c_void_p), ("mat_id", C.c_void_p), ("tex_ids", C.c_void_p),
                ("pos", C.c_void_p)]


def pack_sphere_set(scene):
    """scene["spheres"] (n,4) centre+radius, scene["sphere_mat_id"] (n,), optional scene["sphere_tex_ids"] (n,4),
    scene["sphere_pos"] (n,) -> (TutuSphereSet or None, objects to keep alive)"""
    if scene.get("spheres") is None or len(scene["spheres"]) == 0:
        return None, []
    sp = np.ascontiguousarray(scene["spheres"], dtype=np.float32).reshape(-1, 4)
    mid = np.ascontiguousarray(scene["sphere_mat_id"], dtype=np.int32)
    if len(mid) != len(sp):
        raise ValueError("sphere_mat_id must have one entry per sphere")
    ss = SphereSet()
    ss.n_spheres, ss.spheres, ss.mat_id = len(sp), sp.ctypes.data, mid.ctypes.data
    keep = [sp, mid]
    if scene.get("sphere_tex_ids") is not None:
        t = np.ascontiguousarray(scene["sphere_tex_ids"], dtype=np.int32).reshape(-1, 4)
        ss.tex_ids = t.ctypes.data
        keep.append(t)
    if scene.get("sphere_pos") is not None:
        q = np.ascontiguousarray(scene["sphere_pos"], dtype=np.int32)
        ss.pos = q.ctypes.data
        keep.append(q)
    return ss, keep


class Texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgb", C.c_void_p)]


class TextureSet(C.Structure):
    _fields_ = [("uvs", C.c_void_p), ("tex_ids", C.c_void_p), ("n_maps", C.c_uint32 * 4), ("maps", C.c_void_p * 4)]


def pack_texture_set(scene):
    """scene["uvs"] (n,6), scene["tex_ids"] (n,4), scene["textures"] {"diffuse": [HxWx3 f32,...], "normal": ..,
    "roughness": .., "metallic": ..} -> (TutuTextureSet or None, objects to keep alive)"""
    if scene.get("uvs") is None or scene.get("tex_ids") is None:
        return None, []
    uvs = np.ascontiguousarray(scene["uvs"], dtype=np.float32).reshape(-1, 6)
    ids = np.ascontiguousarray(scene["tex_ids"], dtype=np.int32).reshape(-1, 4)
    n = len(np.asarray(scene["verts"]).reshape(-1, 9))
    if len(uvs) != n or len(ids) != n:
        raise ValueError("uvs / tex_ids must have one row per triangle")
    ts = TextureSet()
    ts.uvs, ts.tex_ids = uvs.ctypes.data, ids.ctypes.data
    keep = [uvs, ids]
    for k, name in enumerate(("diffuse", "normal", "roughness", "metallic")):
        lst = scene.get("textures", {}).get(name, [])
        arr = (Texture * max(1, len(lst)))()
        for i, img in enumerate(lst):
            a = np.ascontiguousarray(img, dtype=np.float32)
            if a.ndim != 3 or a.shape[2] != 3:
                raise ValueError("a texture is an H x W x 3 float array")
            arr[i].height, arr[i].width, arr[i].rgb = a.shape[0], a.shape[1], a.ctypes.data
            keep.append(a)
        ts.n_maps[k] = len(lst)
        ts.maps[k] = C.cast(arr, C.c_void_p).value
        keep.append(arr)
    return ts, keep


class CameraDesc(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("eye", C.c_float * 3), ("viewdir", C.c_float * 3),
                ("updir", C.c_float * 3), ("hfov", C.c_int32)]


class CameraFrame(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("ul", C.c_float * 3), ("delta_h", C.c_float * 3),
                ("delta_v", C.c_float * 3), ("c_off_h", C.c_float * 3), ("c_off_v", C.c_float * 3), ("eye", C.c_float * 3)]


class RenderParams(C.Structure):
    _fields_ = [("spp", C.c_int32), ("key0", C.c_uint32), ("key1", C.c_uint32), ("pixels", C.c_void_p),
                ("n_pixels", C.c_int32), ("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32),
                ("spp_per_pass", C.c_int32), ("max_paths", C.c_int64)]


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("segments", C.c_uint64),
                ("passes", C.c_uint32), ("trace_launches", C.c_uint32), ("ms_total", C.c_float), ("ms_trace_closest", C.c_float),
                ("ms_trace_any", C.c_float), ("ms_shade", C.c_float), ("ms_other", C.c_float), ("ms_shade_first", C.c_float),
                ("ms_shade_material", C.c_float), ("ms_shade_terminal", C.c_float), ("shade_material_launches", C.c_uint32),
                ("nodes_closest", C.c_uint64), ("leaves_closest", C.c_uint64), ("nodes_any", C.c_uint64), ("leaves_any", C.c_uint64),
                ("spp_per_pass", C.c_uint32), ("n_sets", C.c_uint32),
                ("wave_node_steps_closest", C.c_uint64), ("wave_leaf_steps_closest", C.c_uint64),
                ("wave_node_steps_any", C.c_uint64), ("wave_leaf_steps_any", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class BvhInfo(C.Structure):
    _fields_ = [("n_tris", C.c_uint32), ("n_inner", C.c_uint32), ("depth", C.c_uint32), ("root_bounds", C.c_float * 6)]


HIT_DTYPE = np.dtype([("t", "<f4"), ("b1", "<f4"), ("b2", "<f4"), ("tri", "<i4")])

_lib = None


def load_library():
    """dlopen the product library and check that it exports the whole ABI.  Raises if it is missing: there is
    no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TutuError(f"{LIB_PATH} is not built -- run `make -C tuturenderer_amd/csrc` (hipcc, gfx950). "
                        "There is no CPU fallback for the render path.")
    lib = C.CDLL(LIB_PATH)
    missing = [s for s in ABI_SYMBOLS if not hasattr(lib, s)]
    if missing and "TUTU_HIP_LIB" in os.environ and all(s in ("tutu_hip_work_ready", "tutu_hip_render_multi_device", "tutu_host_wide8") for s in missing):
        missing = []  # an older build loaded for a same-box A/B run (profiles/ab.sh): the round-4 entry points are not called there
    if missing:
        raise TutuError(f"{LIB_PATH} lacks ABI symbols: {missing}")
    lib.tutu_hip_error_string.restype = C.c_char_p
    lib.tutu_hip_last_error.restype = C.c_char_p
    lib.tutu_hip_version.restype = C.c_char_p
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        lib = load_library()
        msg = lib.tutu_hip_error_string(rc).decode()
        detail = lib.tutu_hip_last_error().decode()
        raise TutuError(f"{what}: {msg} ({rc}) {detail}")


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def device_count():
    lib = load_library()
    n = C.c_int(0)
    rc = lib.tutu_hip_device_count(C.byref(n))
    return n.value if rc == 0 else 0


class CameraRaster(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("fwdDir", C.c_float * 3), ("width", C.c_int32), ("height", C.c_int32),
                ("imagePlaneDist", C.c_float), ("filmPlaneAreaInv", C.c_float), ("lensAreaInv", C.c_float), ("world2raster", C.c_float * 16)]


INTEGRATORS = {"path": 0, "light": 1, "naivept": 2, "bdpt": 3}  # PPMGenerator::integrateType (Renderer.hpp:41-49)
MAX_UNIT_EVENTS = 8


def camera_desc(scene):
    cd = CameraDesc()
    cd.width, cd.height, cd.hfov = int(scene["width"]), int(scene["height"]), int(scene["hfov"])
    cd.eye = (C.c_float * 3)(*[float(x) for x in scene["eye"]])
    cd.viewdir = (C.c_float * 3)(*[float(x) for x in scene["viewdir"]])
    cd.updir = (C.c_float * 3)(*[float(x) for x in scene["updir"]])
    return cd


def camera_raster(scene):
    """tutu_camera_raster: what LightTracing / NaivePT / BDPT read from g->cam (Camera.hpp:12-79)."""
    lib = load_library()
    cd = camera_desc(scene)
    cr = CameraRaster()
    _check(lib.tutu_camera_raster(C.byref(cd), C.byref(cr)), "tutu_camera_raster")
    return cr


def camera_frame(scene):
    """tutu_camera_frame: Camera::initialize + the camera-frame lines of PathTracing::integrate."""
    lib = load_library()
    cd = camera_desc(scene)
    cf = CameraFrame()
    _check(lib.tutu_camera_frame(C.byref(cd), C.byref(cf)), "tutu_camera_frame")
    return cf


def camera_frame_array(scene):
    cf = camera_frame(scene)
    return np.array([list(cf.ul), list(cf.delta_h), list(cf.delta_v), list(cf.c_off_h), list(cf.c_off_v), list(cf.eye)], np.float32)


def bvh_build_preorder(verts):
    """Host BVH build (same tree as BVHAccel::recursiveBuild); returns (bounds6, leaf_tri, info dict)."""
    lib = load_library()
    verts = _f32(verts).reshape(-1, 9)
    n = len(verts)
    cap = max(1, 2 * n)
    b = np.zeros((cap, 6), np.float32)
    leaf = np.zeros(cap, np.int32)
    nn = C.c_uint32(0)
    info = BvhInfo()
    _check(lib.tutu_bvh_build_preorder(C.c_uint32(n), _p(verts), C.c_uint32(cap), C.byref(nn), _p(b), _p(leaf), C.byref(info)),
           "tutu_bvh_build_preorder")
    return b[: nn.value].copy(), leaf[: nn.value].copy(), {"n_tris": info.n_tris, "n_inner": info.n_inner, "depth": info.depth,
                                                            "root_bounds": np.array(list(info.root_bounds), np.float32)}


def scene_desc(scene):
    """(SceneDesc, keep-alive list) for a scene dict as produced by tuturenderer_amd.scenes"""
    verts = _f32(scene["verts"]).reshape(-1, 9)
    normals = _f32(scene["normals"]).reshape(-1, 9)
    mat_id = np.ascontiguousarray(scene["mat_id"], dtype=np.int32)
    mats = np.ascontiguousarray(scene["mats"], dtype=MAT_DTYPE)
    d = SceneDesc()
    d.n_tris = len(verts)
    d.verts, d.normals, d.mat_id = verts.ctypes.data, normals.ctypes.data, mat_id.ctypes.data
    d.n_mats = len(mats)
    d.mats = mats.ctypes.data
    d.eta = float(scene.get("eta", 1.0))
    d.bkg = (C.c_float * 3)(*[float(x) for x in scene.get("bkg", (0, 0, 0))])
    tex, tex_keep = pack_texture_set(scene)
    d.textures = C.addressof(tex) if tex is not None else None
    sph, sph_keep = pack_sphere_set(scene)
    d.spheres = C.addressof(sph) if sph is not None else None
    return d, [verts, normals, mat_id, mats, tex, tex_keep, sph, sph_keep]


def host_wide8(scene):
    """tutu_host_wide8 (no GPU): the eight-wide tree the library builds for `scene` -> (nodes as an (n_ids, 32) uint32 array,
    info dict with n_nodes / depth / margin, the reference's leaf boxes (n_objects, 8))"""
    lib = load_library()
    d, keep = scene_desc(scene)
    n_ids, n_nodes, depth, n_obj, margin = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0), C.c_uint32(0), C.c_double(0)
    _check(lib.tutu_host_wide8(C.byref(d), C.c_uint32(0), None, C.byref(n_ids), C.byref(n_nodes), C.byref(depth), C.c_uint32(0), None, C.byref(n_obj),
                               C.byref(margin)), "tutu_host_wide8")
    nodes = np.zeros((n_ids.value, 32), np.uint32)
    boxes = np.zeros((n_obj.value, 8), np.float32)
    _check(lib.tutu_host_wide8(C.byref(d), n_ids, _p(nodes), C.byref(n_ids), C.byref(n_nodes), C.byref(depth), n_obj, _p(boxes), C.byref(n_obj),
                               C.byref(margin)), "tutu_host_wide8")
    del keep
    return nodes, {"n_nodes": n_nodes.value, "depth": depth.value, "margin": margin.value}, boxes


class Context:
    """One scene on one GPU (TutuCtx).  `scene` is a dict as produced by tuturenderer_amd.scenes."""

    def __init__(self, scene, device=0):
        self.lib = load_library()
        self.scene = scene
        self.W, self.H = int(scene["width"]), int(scene["height"])
        self._verts = _f32(scene["verts"]).reshape(-1, 9)
        self._normals = _f32(scene["normals"]).reshape(-1, 9)
        self._mat_id = np.ascontiguousarray(scene["mat_id"], dtype=np.int32)
        self._mats = np.ascontiguousarray(scene["mats"], dtype=MAT_DTYPE)
        d = SceneDesc()
        d.n_tris = len(self._verts)
        d.verts, d.normals, d.mat_id = self._verts.ctypes.data, self._normals.ctypes.data, self._mat_id.ctypes.data
        d.n_mats = len(self._mats)
        d.mats = self._mats.ctypes.data
        d.eta = float(scene.get("eta", 1.0))
        d.bkg = (C.c_float * 3)(*[float(x) for x in scene.get("bkg", (0, 0, 0))])
        self._texture_set, self._texture_keep = pack_texture_set(scene)
        d.textures = C.addressof(self._texture_set) if self._texture_set is not None else None
        self._sphere_set, self._sphere_keep = pack_sphere_set(scene)
        d.spheres = C.addressof(self._sphere_set) if self._sphere_set is not None else None
        self.h = C.c_void_p()
        _check(self.lib.tutu_hip_create(C.byref(d), C.c_int(device), C.byref(self.h)), "tutu_hip_create")
        self.cam = camera_frame(scene)
        self.cam_desc = camera_desc(scene)
        self.device = device
        self.last_stats = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.tutu_hip_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def postprocess(self, img, stage=0):
        """Postprocessor.hpp on the device: 0 = performPostProcess (HDR_BLOOM), 1 = emissive, 2 = Gaussian blur, 3 = tone map"""
        img = np.ascontiguousarray(img, dtype=np.float32)
        out = np.empty_like(img)
        _check(self.lib.tutu_hip_postprocess(self.h, C.c_int32(stage), C.c_int32(img.shape[1]), C.c_int32(img.shape[0]), _p(img), _p(out)),
               "tutu_hip_postprocess")
        return out

    def quantise(self, values):
        v = _f32(values)
        out = np.empty(v.shape, np.int32)
        _check(self.lib.tutu_hip_quantise(self.h, C.c_uint32(v.size), _p(v), _p(out)), "tutu_hip_quantise")
        return out

    def set_option(self, name, value):
        _check(self.lib.tutu_hip_set_option(self.h, C.c_char_p(name.encode()), C.c_int(int(value))), "tutu_hip_set_option")

    OPTION_NAMES = ("sets", "sets_default", "one_set", "shade_bpc", "trace_bpc", "refill_min", "inner_steps", "inner_steps_any", "leaf_again", "trace_xcd", "kernel_events", "any_near_first",
                    "util_stats", "bidir_units", "sah_tree", "n_refs", "lds_scene", "shade_tab", "fast_depth", "stack_entries", "stack_entries_hbm", "lds_stack_max", "wide", "wide_min_mb", "wide_inner_steps", "wide_inner_steps_any", "wide_tree", "wide_depth", "wide_early", "pair_leaves",
                    "trace_blocks_per_cu", "trace_lds_bytes", "wide_lds_stack", "wide_early_max_mb", "exact", "exact_sum", "trace_deal", "wide8_top", "wide8_top_nodes", "cold_paths_mi", "work_paths_mi", "device_build", "device_build_min_k", "device_built", "flat", "flat_leaves", "flat_share", "wide_greedy",
                    "paths_mi", "gather_rccl", "gather_path", "peer_access", "rccl_available", "last_trace_us",
                    "wide8", "wide8_tree", "wide8_depth", "wide8_nodes", "wide8_entries", "wide8_inner_steps", "wide8_inner_steps_any", "wide8_leaf_steps", "wide8_leaf_again", "wide8_leaf_room")

    def work_ready(self, wait=False):
        """tutu_hip_work_ready: False while a background thread still allocates this context's full-size work sets (cold start)"""
        rc = self.lib.tutu_hip_work_ready(self.h, C.c_int(1 if wait else 0))
        if rc < 0:
            _check(rc, "tutu_hip_work_ready")
        return rc == 0

    def get_option(self, name):
        v = C.c_int(0)
        _check(self.lib.tutu_hip_get_option(self.h, C.c_char_p(name.encode()), C.byref(v)), "tutu_hip_get_option")
        return v.value

    def options(self):
        """effective value of every knob (what a benchmark line should echo)"""
        out = {}
        for k in self.OPTION_NAMES:
            try:
                out[k] = self.get_option(k)
            except TutuError:  # an older build loaded through TUTU_HIP_LIB (same-box A/B runs, profiles/ab.sh)
                out[k] = None
        return out

    def info(self):
        b = BvhInfo()
        nl = C.c_uint32(0)
        _check(self.lib.tutu_hip_scene_info(self.h, C.byref(b), C.byref(nl)), "tutu_hip_scene_info")
        return {"n_tris": b.n_tris, "n_inner": b.n_inner, "depth": b.depth, "n_lights": nl.value}

    def _params(self, spp, key0, key1, pixels, rect, spp_per_pass, max_paths):
        rp = RenderParams()
        rp.spp, rp.key0, rp.key1 = int(spp), int(key0), int(key1)
        rp.spp_per_pass, rp.max_paths = int(spp_per_pass), int(max_paths)
        keep = None
        if pixels is not None:
            keep = np.ascontiguousarray(pixels, dtype=np.int32)
            rp.pixels, rp.n_pixels = keep.ctypes.data, len(keep)
            n = len(keep)
        else:
            if rect is None:
                rect = (0, 0, self.W, self.H)
            rp.x0, rp.y0, rp.x1, rp.y1 = [int(v) for v in rect]
            n = (rp.x1 - rp.x0) * (rp.y1 - rp.y0)
        return rp, n, keep, rect

    def render(self, spp, key0, key1, pixels=None, rect=None, spp_per_pass=0, max_paths=0, full_frame=True):
        """IIntegrator::integrate.  Returns (H,W,3) float32 when full_frame (untouched pixels are 0), else the
        compact (n,3) array of the work items."""
        rp, n, keep, rect = self._params(spp, key0, key1, pixels, rect, spp_per_pass, max_paths)
        out = np.zeros((n, 3), np.float32)
        st = Stats()
        _check(self.lib.tutu_hip_render(self.h, C.byref(self.cam), C.byref(rp), _p(out), C.byref(st)), "tutu_hip_render")
        self.last_stats = st.as_dict()
        if not full_frame:
            return out
        img = np.zeros((self.H * self.W, 3), np.float32)
        if keep is not None:
            img[keep] = out
        else:
            x0, y0, x1, y1 = rect
            img.reshape(self.H, self.W, 3)[y0:y1, x0:x1] = out.reshape(y1 - y0, x1 - x0, 3)
        return img.reshape(self.H, self.W, 3)

    @staticmethod
    def render_multi(ctxs, spp, key0, key1, pixels=None, rect=None, spp_per_pass=0, max_paths=0):
        """tutu_hip_render_multi over several contexts of the same scene; returns the compact (n,3) array"""
        c0 = ctxs[0]
        rp, n, keep, rect = c0._params(spp, key0, key1, pixels, rect, spp_per_pass, max_paths)
        out = np.zeros((n, 3), np.float32)
        arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
        st = (Stats * len(ctxs))()
        _check(c0.lib.tutu_hip_render_multi(arr, C.c_int32(len(ctxs)), C.byref(c0.cam), C.byref(rp), _p(out), st), "tutu_hip_render_multi")
        for c, s_ in zip(ctxs, st):
            c.last_stats = s_.as_dict()
        return out

    @staticmethod
    def render_multi_device(ctxs, d_out_ptr, spp, key0, key1, pixels=None, rect=None, spp_per_pass=0, max_paths=0, stream=None):
        """tutu_hip_render_multi_device: the frame (compact (n,3) rows in work-item order) is left in device memory of ctxs[0]'s
        device at d_out_ptr; the pieces are gathered there by peer copies and one un-tiling kernel"""
        c0 = ctxs[0]
        rp, n, keep, rect = c0._params(spp, key0, key1, pixels, rect, spp_per_pass, max_paths)
        arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
        st = (Stats * len(ctxs))()
        _check(c0.lib.tutu_hip_render_multi_device(arr, C.c_int32(len(ctxs)), C.byref(c0.cam), C.byref(rp), C.c_void_p(int(d_out_ptr)),
                                                   C.c_void_p(int(stream)) if stream else None, st), "tutu_hip_render_multi_device")
        for c, s_ in zip(ctxs, st):
            c.last_stats = s_.as_dict()
        return n

    def render_device(self, d_out_ptr, spp, key0, key1, pixels=None, rect=None, spp_per_pass=0, max_paths=0, stream=None):
        """Same, writing the compact (n,3) result to device memory at d_out_ptr (e.g. torch_tensor.data_ptr())."""
        rp, n, keep, rect = self._params(spp, key0, key1, pixels, rect, spp_per_pass, max_paths)
        st = Stats()
        _check(self.lib.tutu_hip_render_device(self.h, C.byref(self.cam), C.byref(rp), C.c_void_p(d_out_ptr),
                                               C.c_void_p(stream) if stream else None, C.byref(st)), "tutu_hip_render_device")
        self.last_stats = st.as_dict()
        return n

    def trace_samples(self, pix, smp, key0, key1):
        pix = np.ascontiguousarray(pix, dtype=np.uint32)
        smp = np.ascontiguousarray(smp, dtype=np.uint32)
        L = np.zeros((len(pix), 3), np.float32)
        _check(self.lib.tutu_hip_trace_samples(self.h, C.byref(self.cam), C.c_uint32(len(pix)), _p(pix), _p(smp), C.c_uint32(key0),
                                               C.c_uint32(key1), _p(L)), "tutu_hip_trace_samples")
        return L

    # ---- the other integrators behind the seam (LightTracing / NaivePT / BDPT)
    def render_integrator(self, integrator, spp, key0, key1):
        """IIntegrator::integrate of LightTracing / NaivePT / BDPT over the whole frame -> (H, W, 3) linear radiance."""
        t = INTEGRATORS[integrator] if isinstance(integrator, str) else int(integrator)
        out = np.zeros((self.H, self.W, 3), np.float32)
        st = Stats()
        _check(self.lib.tutu_hip_render_integrator(self.h, C.c_int32(t), C.byref(self.cam_desc), C.c_int32(int(spp)), C.c_uint32(key0), C.c_uint32(key1),
                                                   _p(out), C.byref(st)), "tutu_hip_render_integrator")
        self.last_stats = st.as_dict()
        return out

    def integrator_samples(self, integrator, spp, pix, smp, key0, key1):
        """Units (pixel, sample) of those integrators: dict(own (n,3), alive (n,), n_ev (n,), ev_op / ev_index (n, 8), ev_rgb (n, 8, 3))."""
        t = INTEGRATORS[integrator] if isinstance(integrator, str) else int(integrator)
        pix = np.ascontiguousarray(pix, dtype=np.uint32)
        smp = np.ascontiguousarray(smp, dtype=np.uint32)
        n, m = len(pix), MAX_UNIT_EVENTS
        own = np.zeros((n, 3), np.float32)
        alive = np.zeros(n, np.uint8)
        n_ev = np.zeros(n, np.int32)
        ev_op = np.full((n, m), -1, np.int32)
        ev_index = np.full((n, m), -1, np.int32)
        ev_rgb = np.zeros((n, m, 3), np.float32)
        _check(self.lib.tutu_hip_integrator_samples(self.h, C.c_int32(t), C.byref(self.cam_desc), C.c_int32(int(spp)), C.c_uint32(n), _p(pix), _p(smp),
                                                    C.c_uint32(key0), C.c_uint32(key1), _p(own), _p(alive), C.c_int32(m), _p(n_ev), _p(ev_op), _p(ev_index),
                                                    _p(ev_rgb)), "tutu_hip_integrator_samples")
        return dict(own=own, alive=alive, n_ev=n_ev, ev_op=ev_op, ev_index=ev_index, ev_rgb=ev_rgb)

    # ---- kernel-level entry points
    def trace_closest(self, o, d):
        o, d = _f32(o), _f32(d)
        hits = np.zeros(len(o), HIT_DTYPE)
        _check(self.lib.tutu_hip_trace_closest(self.h, C.c_uint32(len(o)), _p(o), _p(d), _p(hits)), "tutu_hip_trace_closest")
        return hits

    def trace_any(self, orig, target):
        orig, target = _f32(orig), _f32(target)
        b = np.zeros(len(orig), np.uint8)
        _check(self.lib.tutu_hip_trace_any(self.h, C.c_uint32(len(orig)), _p(orig), _p(target), _p(b)), "tutu_hip_trace_any")
        return b

    def eval_bxdf(self, mat, wi, wo, Ng, Ns, eta_scene=1.0, tir=None):
        wi, wo, Ng, Ns = map(_f32, (wi, wo, Ng, Ns))
        m = np.ascontiguousarray(mat, dtype=MAT_DTYPE)
        out = np.zeros((len(wi), 3), np.float32)
        tirp = None
        if tir is not None:
            tir = np.ascontiguousarray(tir, dtype=np.uint8)
            tirp = _p(tir)
        _check(self.lib.tutu_hip_eval_bxdf(self.h, C.c_uint32(len(wi)), _p(m), _p(wi), _p(wo), _p(Ng), _p(Ns), C.c_float(eta_scene), tirp,
                                           _p(out)), "tutu_hip_eval_bxdf")
        return out

    def eval_pdf(self, mat, wi, wo, N, eta_i=1.0, eta_t=None):
        wi, wo, N = map(_f32, (wi, wo, N))
        m = np.ascontiguousarray(mat, dtype=MAT_DTYPE)
        if eta_t is None:
            eta_t = float(m["eta"])
        out = np.zeros(len(wi), np.float32)
        _check(self.lib.tutu_hip_eval_pdf(self.h, C.c_uint32(len(wi)), _p(m), _p(wi), _p(wo), _p(N), C.c_float(eta_i), C.c_float(eta_t),
                                          _p(out)), "tutu_hip_eval_pdf")
        return out

    def eval_texture(self, list_index, index, u, v):
        u, v = _f32(u), _f32(v)
        out = np.zeros((len(u), 3), np.float32)
        _check(self.lib.tutu_hip_eval_texture(self.h, C.c_int32(list_index), C.c_int32(index), C.c_uint32(len(u)), _p(u), _p(v), _p(out)),
               "tutu_hip_eval_texture")
        return out

    def eval_sample(self, mat, wo, N, xi3, eta_i=1.0):
        wo, N, xi3 = map(_f32, (wo, N, xi3))
        m = np.ascontiguousarray(mat, dtype=MAT_DTYPE)
        n = len(wo)
        wi = np.zeros((n, 3), np.float32)
        ok = np.zeros(n, np.uint8)
        sp = np.zeros(n, np.uint8)
        nd = np.zeros(n, np.int32)
        _check(self.lib.tutu_hip_eval_sample(self.h, C.c_uint32(n), _p(m), _p(wo), _p(N), C.c_float(eta_i), _p(xi3), _p(wi), _p(ok), _p(sp),
                                             _p(nd)), "tutu_hip_eval_sample")
        return wi, ok, sp, nd

    FN = {"bbox": (0, 1), "tri": (1, 11), "normalized": (2, 3), "fresnel": (3, 1), "fresnel_schlick": (4, 3), "reflect": (5, 3),
          "refract": (6, 3), "D": (7, 1), "G": (8, 1), "mis": (9, 1), "local2world": (10, 3), "rng": (11, 8), "philox": (12, 4), "libm": (13, 8)}

    def eval_fn(self, name, *arrays):
        """tutu_hip_eval_fn: one of the hot-path device functions on arrays (float32; the rng / philox inputs are uint32
        rows viewed as float32 bits).  Returns an (n, width) float32 array."""
        fn, w_out = self.FN[name]
        arrs = [np.ascontiguousarray(a).view(np.float32) if np.asarray(a).dtype == np.uint32 else _f32(a) for a in arrays]
        n = len(arrs[0])
        ptrs = (C.c_void_p * 6)(*[a.ctypes.data for a in arrs] + [None] * (6 - len(arrs)))
        out = np.zeros((n, w_out), np.float32)
        _check(self.lib.tutu_hip_eval_fn(self.h, C.c_int32(fn), C.c_uint32(n), ptrs, _p(out)), "tutu_hip_eval_fn")
        return out

    def eval_sample_light(self, xi3):
        xi3 = _f32(xi3)
        n = len(xi3)
        tri = np.zeros(n, np.int32)
        pos = np.zeros((n, 3), np.float32)
        nrm = np.zeros((n, 3), np.float32)
        pdf = np.zeros(n, np.float32)
        _check(self.lib.tutu_hip_eval_sample_light(self.h, C.c_uint32(n), _p(xi3), _p(tri), _p(pos), _p(nrm), _p(pdf)),
               "tutu_hip_eval_sample_light")
        return tri, pos, nrm, pdf
