"""The drop-in boundary, proven against the reference itself.

oracle/Makefile builds, in the build container only (where /root/reference exists), into oracle/_ref/:
  * ref_binding           the reference's OWN front-end (PPMGenerator, objl::Loader, loadObj, the P3 writer -- compiled from
                          /root/reference/include where the sources lie) + tuturenderer_amd/integration/HipPathTracing.hpp,
                          the file a maintainer adds to the reference;
  * main_cornellBox_ref,  the reference's OWN scene programs (src/main_cornellBox.cpp, src/main_veach_bdpt.cpp, compiled where
    main_veach_ref        they lie) over the bundled front-end's headers -- unchanged statements, our API surface.
and, everywhere, oracle/bundled_binding = the ref_binding program over the bundled front-end.

CPU tests (here): the builds succeed, and both front-ends hand the SAME bytes to tutu_hip_create (FNV hash of the flattened
scene, camera frame) -- `--dry`, no device call.  GPU tests: the prebuilt binaries render and write byte-identical PPM files."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from test_host_frontend import APPS, write_obj

REF_BIN = os.path.join(ROOT, "oracle", "_ref")
CONFIG = "imsize 96 72\neye 278 273 -800\nviewdir 0 0 1\nhfov 40\nupdir 0 1 0\nbkgcolor 0 0 0 1.0\nintegrator path"


def _cornell_files(tmp_path, with_model_dir=False):
    """the Cornell meshes as OBJ files (no vn lines, like the reference's) + a config; returns (cfg, mesh specs)"""
    from tuturenderer_amd import scenes

    d = tmp_path / "model" / "cornellBox"
    d.mkdir(parents=True)
    run = tmp_path / "run"
    run.mkdir()
    mats = {"floor": scenes.CB_WHITE, "light": (0.9, 0.9, 0.9), "right": scenes.CB_GREEN, "left": scenes.CB_RED, "tallbox": scenes.CB_WHITE,
            "shortbox": scenes.CB_WHITE}
    specs = []
    for name, v in scenes.cornell_parts():
        write_obj(d / f"{name}.obj", v)
        em = scenes.CB_EMISSION if name == "light" else (0, 0, 0)
        specs.append(f"{d / (name + '.obj')}|0|{','.join(repr(float(np.float32(x))) for x in mats[name])}|{','.join(repr(float(np.float32(x))) for x in em)}|1|1|0")
    cfg = run / "cfg.txt"
    cfg.write_text(CONFIG)
    return cfg, specs, run


def _have_ref_builds():
    return all(os.path.exists(os.path.join(REF_BIN, b)) for b in ("ref_binding", "main_cornellBox_ref", "main_veach_ref"))


def test_reference_builds_exist_where_the_reference_does(built):
    """in the build container the three boundary builds must have been produced by `make -C oracle` (conftest.built)"""
    if not os.path.exists("/root/reference/include/PathTracing.hpp"):
        pytest.skip("no /root/reference here (GPU box): the prebuilt binaries are used")
    assert _have_ref_builds()
    # the scene programs were compiled from the reference's own files (symlinks), not from copies
    for f in ("main_cornellBox.cpp", "main_veach_bdpt.cpp"):
        link = os.path.join(REF_BIN, "mains", "src", f)
        assert os.path.islink(link) and os.path.realpath(link) == os.path.join("/root/reference/src", f)


def test_binding_hands_over_the_same_scene_from_both_front_ends(built, tmp_path):
    """ref_binding --dry (reference front-end) == bundled_binding --dry (bundled front-end): triangle / material / light
    counts, the FNV hash over every byte tutu_hip_create would read, and the camera frame"""
    if not _have_ref_builds():
        pytest.skip("oracle/_ref boundary builds missing (needs /root/reference at build time)")
    cfg, specs, run = _cornell_files(tmp_path)
    outs = []
    for exe in (os.path.join(REF_BIN, "ref_binding"), os.path.join(ROOT, "oracle", "bundled_binding")):
        r = subprocess.run([exe, str(cfg), "--dry"] + specs, cwd=run, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        lines = [l for l in r.stdout.splitlines() if l.startswith(("tris ", "frame "))]
        assert len(lines) == 2, r.stdout
        outs.append(lines)
    assert outs[0] == outs[1], outs
    assert outs[0][0].startswith("tris 32 mats 4 spheres 0 lights 2 hash ")


@pytest.mark.gpu
def test_reference_builds_travelled_to_the_gpu_box():
    """GPU box guard: the reference-built checker binaries (oracle/_ref, git-ignored but NOT gpurun-ignored) must have come
    along with the snapshot -- a push that drops them would otherwise silently shrink the suite to its skips"""
    from oracle.pyoracle import available

    missing = [b for b in ("ref_binding", "main_cornellBox_ref", "main_veach_ref", "libtutu_ref.so", "libtutu_ref_fast.so")
               if not os.path.exists(os.path.join(REF_BIN, b))]
    assert not missing, f"oracle/_ref is incomplete on this box: {missing}"
    assert available("reference") and available("reference_fast")


@pytest.mark.gpu
def test_reference_front_end_renders_through_the_binding(built, tmp_path):
    """GPU: the reference's own front-end + HipPathTracing.hpp renders the Cornell box and writes, with the reference's own
    P3 writer, the same bytes as the bundled front-end does; a second integrate() of the same scene reuses the contexts"""
    assert _have_ref_builds(), "oracle/_ref boundary builds missing on the GPU box"
    cfg, specs, run = _cornell_files(tmp_path)
    ppm = {}
    for tag, exe in (("ref", os.path.join(REF_BIN, "ref_binding")), ("ours", os.path.join(ROOT, "oracle", "bundled_binding"))):
        r = subprocess.run([exe, str(cfg), "--spp", "8", "--key1", "7"] + specs, cwd=run, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        ppm[tag] = open(run / "cfg.ppm", "rb").read()
        os.remove(run / "cfg.ppm")
    assert ppm["ref"][:12] == b"P3\n96\n72\n255"
    assert ppm["ref"] == ppm["ours"]
    # and the picture is the library's: compare with the Python front-end at the same key
    import tuturenderer_amd as tr
    from tuturenderer_amd import scenes

    sc = scenes.cornell_box(96, 72)
    with tr.Context(sc) as ctx:
        img = ctx.render(8, 0x5EED0001, 7)
        lv = ctx.quantise(img.reshape(-1))
    got = np.array(ppm["ref"].split()[4:], np.int32)
    assert (got != lv).mean() < 1e-3  # quantise() on the device vs powf on the host: isolated one-level differences
    # `integrator bdpt` in the config: the reference's front-end hands g->integrateType to the same binding
    bcfg = run / "b.txt"
    bcfg.write_text(CONFIG.replace("integrator path", "integrator bdpt"))
    r = subprocess.run([os.path.join(REF_BIN, "ref_binding"), str(bcfg), "--spp", "4", "--key1", "7"] + specs, cwd=run, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.array(open(run / "b.ppm", "rb").read().split()[4:], np.int32)
    # (the mesh specs give the light the Material default diffuse 0.9, as src/main_cornellBox.cpp does; BDPT connects through
    # light-path vertices that landed on the emitter and reads it there, PathTracing never does)
    emissive = np.flatnonzero(np.asarray(sc["mats"]["emission"]).any(1))
    sc["mats"]["diffuse"][emissive] = np.float32(0.9)
    with tr.Context(sc) as ctx:
        lv = ctx.quantise(ctx.render_integrator("bdpt", 4, 0x5EED0001, 7).reshape(-1))
    assert (got != lv).mean() < 1e-3


@pytest.mark.gpu
def test_reference_scene_programs_over_the_bundled_front_end(built, tmp_path):
    """GPU: src/main_cornellBox.cpp of the reference, compiled unchanged against host/tutu_renderer.hpp, end to end -- the same
    PPM as the bundled scene program apps/main_cornellBox (which restates it)"""
    assert _have_ref_builds(), "oracle/_ref boundary builds missing on the GPU box"
    cfg, specs, run = _cornell_files(tmp_path)
    out = {}
    for tag, exe in (("ref_main", os.path.join(REF_BIN, "main_cornellBox_ref")), ("app", os.path.join(APPS, "main_cornellBox"))):
        # both load "../model/cornellBox/*.obj" relative to the working directory (src/main_cornellBox.cpp:28)
        r = subprocess.run([exe, str(cfg)], cwd=run, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        out[tag] = open(run / "cfg.ppm", "rb").read()
        os.remove(run / "cfg.ppm")
    assert out["ref_main"] == out["app"]
    assert len(out["app"].split()) == 4 + 96 * 72 * 3
    # the other branches of Renderer's switch (Renderer.hpp:44-49): `integrator light | naivept | bdpt` select the device
    # versions of LightTracing / NaivePT / BDPT through the same binding, from both front-ends
    for integ in ("light", "naivept", "bdpt"):
        icfg = run / f"{integ}.txt"
        icfg.write_text(CONFIG.replace("integrator path", f"integrator {integ}"))
        got = {}
        for tag, exe in (("ref_main", os.path.join(REF_BIN, "main_cornellBox_ref")), ("app", os.path.join(APPS, "main_cornellBox"))):
            r = subprocess.run([exe, str(icfg)], cwd=run, capture_output=True, text=True)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
            got[tag] = open(run / f"{integ}.ppm", "rb").read()
            os.remove(run / f"{integ}.ppm")
        assert got["ref_main"] == got["app"]
        levels = np.array(got["app"].split()[4:], np.int32)
        # (NaivePT shows the emitter and nothing else: the emission of the first hit, NaivePT.hpp:158)
        assert len(levels) == 96 * 72 * 3 and (levels > 0).mean() > (0.003 if integ == "naivept" else 0.3) and got["app"] != out["app"]


@pytest.mark.gpu
def test_multi_context_render_is_the_single_context_frame(built):
    """tutu_hip_render_multi: three contexts (all on device 0 here) deal 32x32 tiles round-robin and write one frame --
    bit-identical to a single context's"""
    import tuturenderer_amd as tr
    from tuturenderer_amd import scenes

    sc = scenes.cornell_box(200, 136)  # not a multiple of the tile size
    ctxs = [tr.Context(sc) for _ in range(3)]
    try:
        one = ctxs[0].render(12, 0x5EED0001, 3, full_frame=False)
        multi = tr.Context.render_multi(ctxs, 12, 0x5EED0001, 3)
        assert multi.tobytes() == one.tobytes()
        rng = np.random.default_rng(9)
        pixels = rng.permutation(200 * 136)[:5000].astype(np.int32)
        a = ctxs[0].render(5, 1, 2, pixels=pixels, full_frame=False)
        b = tr.Context.render_multi(ctxs, 5, 1, 2, pixels=pixels)
        assert a.tobytes() == b.tobytes()
        assert all(c.last_stats["samples"] > 0 for c in ctxs)
        # the device-side gather by itself (tutu_hip_render_multi_device): the frame stays in DEVICE memory, pieces gathered by peer
        # copies in context order, ONE un-tiling kernel, on a stream of the caller's.  (Device memory and the stream come straight
        # from the HIP runtime the library itself is linked with: torch brings a runtime of its own, and the second runtime to
        # initialise in a process finds no GPU.)
        import ctypes as C

        hip = C.CDLL("libamdhip64.so")
        d, d5, stream = C.c_void_p(), C.c_void_p(), C.c_void_p()
        assert hip.hipMalloc(C.byref(d), C.c_size_t(200 * 136 * 12)) == 0 and hip.hipMalloc(C.byref(d5), C.c_size_t(5000 * 12)) == 0
        assert hip.hipStreamCreateWithFlags(C.byref(stream), C.c_uint(1)) == 0  # hipStreamNonBlocking
        try:
            assert hip.hipMemset(d, 0xFF, C.c_size_t(200 * 136 * 12)) == 0 and hip.hipDeviceSynchronize() == 0
            n = tr.Context.render_multi_device(ctxs, d.value, 12, 0x5EED0001, 3, stream=stream.value)
            got = np.empty((200 * 136, 3), np.float32)
            assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), d, C.c_size_t(got.nbytes), C.c_int(2)) == 0  # hipMemcpyDeviceToHost
            assert n == 200 * 136 and got.tobytes() == one.tobytes()
            tr.Context.render_multi_device(ctxs, d5.value, 5, 1, 2, pixels=pixels)
            got5 = np.empty((5000, 3), np.float32)
            assert hip.hipMemcpy(got5.ctypes.data_as(C.c_void_p), d5, C.c_size_t(got5.nbytes), C.c_int(2)) == 0
            assert got5.tobytes() == a.tobytes()
        finally:
            hip.hipFree(d)
            hip.hipFree(d5)
            hip.hipStreamDestroy(stream)
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.gpu
def test_multi_context_gather_through_rccl_is_the_copy_gather(built):
    """tutu_hip_render_multi_device's RCCL leg (round 5; csrc/rccl_gather.cpp): ncclCommInitAll over the contexts' distinct
    devices and ONE ncclGroupStart ... ncclSend / ncclRecv ... ncclGroupEnd per frame into the first context's buffer -- what
    SURVEY.md 8e names, in the C library itself (PathTracing.hpp:393-429 is what it replaces).  What one GPU allows: a
    communicator of ONE rank, every piece by a self send / recv (knob gather_rccl = 2) -- the frame is the copy gather's and
    the single context's, bit for bit, and the library says which path it took.  More than one device has executed nowhere."""
    import tuturenderer_amd as tr
    from tuturenderer_amd import scenes

    sc = scenes.cornell_box(200, 136)
    ctxs = [tr.Context(sc) for _ in range(3)]
    try:
        assert ctxs[0].get_option("rccl_available") == 1, "librccl.so.1 was not found by the library"
        assert ctxs[0].get_option("gather_path") == -1 and ctxs[0].get_option("gather_rccl") == 1
        one = ctxs[0].render(12, 0x5EED0001, 3, full_frame=False)
        by_copies = tr.Context.render_multi(ctxs, 12, 0x5EED0001, 3)
        assert ctxs[0].get_option("gather_path") == 0  # one device: the default keeps the local copies
        ctxs[0].set_option("gather_rccl", 2)
        by_rccl = tr.Context.render_multi(ctxs, 12, 0x5EED0001, 3)
        assert ctxs[0].get_option("gather_path") == 1
        again = tr.Context.render_multi(ctxs, 7, 3, 4)  # the communicator is kept for the next frame
        assert ctxs[0].get_option("gather_path") == 1
        assert by_rccl.tobytes() == one.tobytes() and by_copies.tobytes() == one.tobytes()
        assert again.tobytes() == ctxs[1].render(7, 3, 4, full_frame=False).tobytes()
        ctxs[0].set_option("gather_rccl", 0)
        tr.Context.render_multi(ctxs, 3, 3, 4)
        assert ctxs[0].get_option("gather_path") == 0
    finally:
        for c in ctxs:
            c.close()
