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
    if n.value == 0:
        # no device: creating a context must fail loudly, never fall back to a CPU path
        from tuturenderer_amd import scenes

        with pytest.raises(tr.TutuError):
            tr.Context(scenes.cornell_box(8, 8))
