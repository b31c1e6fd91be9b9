"""CPU: the product library loads, exports the whole C ABI, and its host-side logic (camera frame, BVH build,
argument checking) matches the golden vectors from the reference build.  No compute call needs a GPU here."""
import ctypes as C
import re

import numpy as np
import pytest

from conftest import ROOT, golden_path
from helpers import bit_equal


@pytest.fixture(scope="module")
def tr(built):
    import tuturenderer_amd

    tuturenderer_amd.load_library()
    return tuturenderer_amd


def test_exports_every_declared_symbol(tr):
    hdr = open(f"{ROOT}/include/tutu_hip.h").read()
    declared = set(re.findall(r"\b(tutu_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(tr.ABI_SYMBOLS), declared ^ set(tr.ABI_SYMBOLS)
    lib = tr.load_library()
    for s in declared:
        assert hasattr(lib, s), s
    assert b"gfx950" in lib.tutu_hip_version()


def test_struct_sizes(tr):
    assert tr.MAT_DTYPE.itemsize == 56  # == sizeof(reference Material)
    assert C.sizeof(tr.CameraFrame) == 8 + 6 * 12
    assert tr.HIT_DTYPE.itemsize == 16


SCENES = ["cornell", "cornell_ggxT_mirror", "cornell_ggxR_glass", "veach", "veach_slight", "cornell_degenerate"]


@pytest.mark.parametrize("name", SCENES)
def test_camera_frame_and_bvh_match_reference(tr, name):
    from oracle.gen_golden import golden_scenes

    mk, _ = golden_scenes()[name]
    sc = mk()
    z = np.load(golden_path(f"scene_{name}.npz"))
    assert bit_equal(tr.camera_frame_array(sc), z["scene.camera"])
    b, leaf, info = tr.bvh_build_preorder(sc["verts"])
    assert bit_equal(b, z["bvh.bounds"])
    assert bit_equal(leaf, z["bvh.leaf_tri"])
    assert info["n_inner"] == len(sc["verts"]) - 1
    assert info["depth"] <= 30


@pytest.mark.parametrize("name", SCENES)
def test_camera_raster_matches_the_restatement(tr, port, name):
    """tutu_camera_raster (Camera::initialize's world2Raster and scalars, what the other integrators read) == the CPU
    restatement's, bit for bit -- which equals the reference's own Camera (tests/test_oracle_vs_reference.py)"""
    from oracle.gen_golden import golden_scenes

    mk, _ = golden_scenes()[name]
    sc = mk()
    cr = tr.camera_raster(sc)
    mine = np.array(list(cr.world2raster) + [cr.imagePlaneDist, cr.filmPlaneAreaInv, cr.lensAreaInv] + list(cr.fwdDir), np.float32)
    S = port.scene(sc)
    want = S.camera_raster()
    S.close()
    assert bit_equal(mine, want)
    assert cr.width == sc["width"] and cr.height == sc["height"] and list(cr.position) == [np.float32(x) for x in sc["eye"]]


def test_bvh_edge_cases(tr):
    b, leaf, info = tr.bvh_build_preorder(np.zeros((0, 9), np.float32))
    assert len(leaf) == 0 and info["n_inner"] == 0
    one = np.arange(9, dtype=np.float32).reshape(1, 9)
    b, leaf, info = tr.bvh_build_preorder(one)
    assert list(leaf) == [0] and info["n_inner"] == 0
    two = np.concatenate([one, one + 10])
    b, leaf, info = tr.bvh_build_preorder(two)
    assert list(leaf) == [-1, 0, 1] and info["n_inner"] == 1 and info["depth"] == 1
    # identical centroids: the split must still terminate and keep every triangle exactly once
    many = np.repeat(one, 37, axis=0)
    b, leaf, info = tr.bvh_build_preorder(many)
    assert sorted(leaf[leaf >= 0]) == list(range(37))


def test_argument_errors_without_gpu(tr):
    lib = tr.load_library()
    assert lib.tutu_camera_frame(None, None) == -1
    n = C.c_int(-5)
    lib.tutu_hip_device_count(C.byref(n))
    assert n.value >= 0
    h = C.c_void_p()
    assert lib.tutu_hip_create(None, 0, C.byref(h)) == -1
    assert lib.tutu_hip_error_string(-5).decode().startswith("BVH")
    # the entry points added for textures / spheres / post-processing validate their arguments before any GPU call
    assert lib.tutu_hip_set_option(None, b"sets", 1) == -1
    assert lib.tutu_hip_postprocess(None, 0, 4, 4, None, None) == -1
    assert lib.tutu_hip_quantise(None, 4, None, None) == -1
    assert lib.tutu_hip_eval_texture(None, 0, 0, 4, None, None, None) == -1
    if n.value == 0:
        # no device: creating a context must fail loudly, never fall back to a CPU path
        from tuturenderer_amd import scenes

        with pytest.raises(tr.TutuError):
            tr.Context(scenes.cornell_box(8, 8))


def test_committed_bench_line_keeps_the_contract():
    """the newest bench line under profiles/ (written by bench.py on the GPU box) has every field the driver and the
    measurement contract read, and its roofline is internally consistent"""
    import glob
    import json
    import os

    from conftest import ROOT

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_line_c2.json")))
    assert files, "no committed bench line"
    d = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Msamples/s" and d["higher_is_better"] is True and d["scaling"] == "strong" and d["data"] == "synthetic"
    assert d["dtype"] == "f32" and d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 800 * 800 * 512 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "algorithmic_bytes_per_launch"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    # every kernel has its own entry with its bound; the exclusive (one pass in flight) figures are the top-level ones
    assert set(r["per_kernel"]) == {"k_trace_closest", "k_trace_any", "k_shade"} and r["kernel"] in r["per_kernel"]
    assert abs(r["per_kernel"][r["kernel"]]["frac"] - r["frac"]) < 1e-12 and "exclusive_kernel_ms_per_step" in r
    # (round 5: the top-level kernel is the one with the most exclusive time, whoever that is; k_shade's figures are always under per_kernel)
    assert r["lds_scene"] is True and r["kernel"] == r.get("dominant_by_time", "k_shade") and r["per_kernel"]["k_trace_closest"]["bound"] == "valu"
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["unit"] == "Msamples/s" and c["cores"] >= 1


def test_bench_busy_probe_and_configs_without_a_gpu():
    """bench.py's helpers that need no device: the five BASELINE configs are named, and the sysfs busy probe (round 5: a cross-check
    for a driver-side sampler that polls too slowly) returns a summary whether or not the host has amdgpu cards"""
    import importlib.util
    import os
    import time

    from conftest import ROOT

    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cfg = bench.configs()
    assert sorted(cfg) == ["c1", "c2", "c3", "c4", "c5"] and cfg["c2"]["spp"] == 512 and "BASELINE configs[1]" in cfg["c2"]["name"]
    with bench.BusyProbe() as p:
        time.sleep(0.12)
    s = p.summary()
    assert "samples" in s and (s["samples"] == 0 or 0 <= s["max_percent"] <= 100)


def test_every_python_option_name_is_one_the_library_knows():
    """tuturenderer_amd.Context.options() asks tutu_hip_get_option for every name in OPTION_NAMES (on a GPU box an unknown name raises);
    the same check without a GPU, against the library's source: the knob table and the read-only names of tutu_hip_get_option."""
    import os
    import re

    import tuturenderer_amd

    src = open(os.path.join(os.path.dirname(tuturenderer_amd.__file__), "csrc", "tutu_hip.hip")).read()
    known = set(re.findall(r'\{"([a-z0-9_]+)", "TUTU_[A-Z0-9_]+", &TutuCtx::Knobs::', src)) | set(re.findall(r'strcmp\(name, "([a-z0-9_]+)"\) == 0', src))
    missing = [n for n in tuturenderer_amd.Context.OPTION_NAMES if n not in known]
    assert not missing, missing
    for new in ("exact_sum", "trace_deal", "wide8_top", "wide8_top_nodes", "last_trace_us"):  # round 5
        assert new in tuturenderer_amd.Context.OPTION_NAMES and new in known
