"""CPU, build container only: the port against the live reference build on fresh seeds (skipped where
oracle/_ref/libtutu_ref.so does not exist)."""
import numpy as np

from helpers import assert_dict_bit_equal
from oracle import parity_cases as pc


def test_kind(reference, port):
    assert reference.kind == "reference" and port.kind == "port"


def test_live_samples_other_seed(reference, port):
    from tuturenderer_amd import scenes

    sc = scenes.cornell_box(64, 64)
    R, P = reference.scene(sc), port.scene(sc)
    for key1 in (11, 12):
        assert_dict_bit_equal(pc.run_samples(P, key1, n=800), pc.run_samples(R, key1, n=800))
    a = R.render(4, pc.KEY0, 77, nthreads=4)
    b = P.render(4, pc.KEY0, 77, nthreads=4)
    assert a.tobytes() == b.tobytes()
    R.close()
    P.close()


def test_live_veach_scene_rays(reference, port):
    from tuturenderer_amd import scenes

    sc = scenes.veach_room(64, 48)
    R, P = reference.scene(sc), port.scene(sc)
    assert_dict_bit_equal(pc.run_scene(P, seed=205), pc.run_scene(R, seed=205))
    R.close()
    P.close()


def test_live_other_integrators(reference, port):
    """LightTracing / NaivePT / BDPT on a fresh key: the port == the reference's own integrators, frame for frame"""
    from tuturenderer_amd import scenes

    sc = scenes.veach_room(32, 24, small_light=True)
    R, P = reference.scene(sc), port.scene(sc)
    # the camera the three integrators read: the reference's own Camera after initialize() (world2Raster, Camera.hpp:29-41)
    assert R.camera_raster().tobytes() == P.camera_raster().tobytes()
    for itype in (1, 2, 3):
        a = R.render_integrator(itype, 3, pc.KEY0, 900 + itype)
        b = P.render_integrator(itype, 3, pc.KEY0, 900 + itype)
        assert a.tobytes() == b.tobytes(), itype
    R.close()
    P.close()
    for mk in (lambda: scenes.cornell_box(96, 64), lambda: scenes.cornell_box(33, 57), lambda: scenes.veach_room(120, 90)):
        sc = mk()
        R, P = reference.scene(sc), port.scene(sc)
        assert R.camera_raster().tobytes() == P.camera_raster().tobytes()
        R.close()
        P.close()
