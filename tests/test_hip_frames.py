"""GPU (MI355X): the BASELINE frames, WHOLE, against frames the reference's own code rendered (tests/golden/frame_*.npz,
written by oracle/gen_frames.py from oracle/_ref/libtutu_ref.so = /root/reference/include compiled where it lies, consuming
the same Philox streams through the engine swap of oracle/ref_shim.h): what PathTracing.hpp:485-516 leaves in
g->cam.FrameBuffer.rgb, pixel for pixel.

north_star's bar is: mean over pixels of the per-pixel L2 distance < 1e-3 at matched seed.  The bar HERE since round 5 (the C
library's sinf / cosf / acosf / tanf / powf restated for the device, csrc/device_libm.h) is 1e-6 for that mean, no pixel beyond
1e-4 and frame means equal to 1e-6 -- measured 3e-9 .. 1e-8, worst pixel 4e-6 of a value of 19.5.  Until then 0.1-1.4 % of the
pixels held a path that a last-bit difference in one of those functions had sent another way (mean 1e-6 .. 4e-4).  What is
left -- about half of the pixels differ in a last bit -- is the order of the radiance sum (tests/test_hip_parity.py docstring).

Also here: the statistical pin against the reference's NATIVE random numbers (std::mt19937 per thread, global.hpp:182-199; no
engine swap): frame_native_cornell.npz, 128 x 128 at 4096 spp."""
import numpy as np
import pytest

from conftest import golden_path
from oracle import parity_cases as pc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tr(built):
    import tuturenderer_amd

    tuturenderer_amd.load_library()
    assert tuturenderer_amd.device_count() >= 1, "no HIP device: the product path has no fallback"
    return tuturenderer_amd


def _compare(tag, frame, want):
    assert frame.shape == want.shape and frame.dtype == np.float32
    assert np.isfinite(frame).all() and np.isfinite(want).all()
    d = np.sqrt(((frame.astype(np.float64) - want.astype(np.float64)) ** 2).sum(-1))
    mean_l2 = float(d.mean())
    share = float((d > 1e-2).mean())
    worst = np.unravel_index(int(d.argmax()), d.shape)
    equal = float((frame.view(np.uint32) == want.view(np.uint32)).all(-1).mean())
    print(f"\n[{tag}] {frame.shape[1]}x{frame.shape[0]}: mean per-pixel L2 vs the reference build {mean_l2:.3e} (bar here 1e-6; north_star 1e-3); pixels beyond 1e-2: "
          f"{share:.3%}; worst pixel (x={worst[1]}, y={worst[0]}) {d.max():.3e} of value {np.abs(want[worst]).max():.3f}; "
          f"bit-equal pixels {equal:.1%}; frame means {frame.mean():.6f} / {want.mean():.6f}")
    assert mean_l2 < 1e-6, mean_l2
    assert share == 0.0 and d.max() < 1e-4 * max(1.0, float(np.abs(want[worst]).max())), (share, float(d.max()))
    assert abs(float(frame.mean(dtype=np.float64)) - float(want.mean(dtype=np.float64))) < 1e-6
    return mean_l2, share


def _frame(tr, name, mk, **env):
    z = np.load(golden_path(f"frame_{name}.npz"))
    sc = mk()
    assert pc.checksum(np.ascontiguousarray(sc["verts"], np.float32), np.ascontiguousarray(sc["normals"], np.float32),
                       np.ascontiguousarray(sc["mat_id"], np.int32)) == z["scene_crc"], "the scene generator drifted from the fixture's"
    with tr.Context(sc) as ctx:
        frame = ctx.render(int(z["spp"]), int(z["key0"]), int(z["key1"]))
    return frame, z["rgb"]


def test_c2_cornell_800x800_512spp_whole_frame_vs_reference_build(tr):
    """BASELINE configs[1] -- the headline frame bench.py times"""
    from tuturenderer_amd import scenes

    frame, want = _frame(tr, "c2", lambda: scenes.cornell_box(800, 800))
    _compare("c2 cornell 512 spp", frame, want)


def test_c1_cornell_800x800_16spp_whole_frame_vs_reference_build(tr):
    """BASELINE configs[0] -- config_cornellBox.txt's own CPU-runnable case at the harness size: 16 spp, key 1 (round 5)"""
    from tuturenderer_amd import scenes

    frame, want = _frame(tr, "c1", lambda: scenes.cornell_box(800, 800))
    _compare("c1 cornell 16 spp", frame, want)


def test_c5_veach_800x600_512spp_whole_frame_vs_reference_build(tr):
    """BASELINE configs[4]: Lambertian + PERFECT_REFRACTIVE + MICROFACET_R, wide tree"""
    from tuturenderer_amd import scenes

    frame, want = _frame(tr, "c5", lambda: scenes.veach_room(800, 600, small_light=False))
    _compare("c5 veach room 512 spp", frame, want)


def test_c3_bunny_stand_in_1024x1024_256spp_whole_frame_vs_reference_build(tr):
    """BASELINE configs[2] on the stand-in mesh (MICROFACET_T blob, 82 k triangles)"""
    from tuturenderer_amd import scenes

    frame, want = _frame(tr, "c3", lambda: scenes.bunny_box(1024, 1024))
    _compare("c3 bunny stand-in 256 spp", frame, want)


def test_c4_broom_stand_in_1600x900_16spp_whole_frame_vs_reference_build(tr, monkeypatch):
    """BASELINE configs[3] on the stand-in (48 k needle-shaped triangles, 665 k CLIPPED references in the walked tree) at 16 of its
    1024 spp -- what the reference's CPU integrator finishes in minutes.  The default walk (clipped references, distance pruning)
    AND the exact walk (TUTU_EXACT=1: the reference's own tree, no pruning) against the reference build's frame; the two device
    frames against each other bit for bit: 23 M samples = 0.19 G rays on which pruning against clipped references lost nothing."""
    from tuturenderer_amd import scenes

    frame, want = _frame(tr, "c4", lambda: scenes.broom_room(1600, 900))
    _compare("c4 broom stand-in 16 spp, default walk", frame, want)
    monkeypatch.setenv("TUTU_EXACT", "1")
    exact, _ = _frame(tr, "c4", lambda: scenes.broom_room(1600, 900))
    monkeypatch.delenv("TUTU_EXACT")
    _compare("c4 broom stand-in 16 spp, exact walk", exact, want)
    n_diff = int((frame.view(np.uint32) != exact.view(np.uint32)).any(-1).sum())
    print(f"[c4] pixels that differ between the default walk and the exact walk: {n_diff}")
    assert n_diff == 0


def test_c4_broom_stand_in_64spp_whole_frame_vs_reference_build(tr):
    """Round 5.  The 16-spp pin of BASELINE configs[3] used to sit 2.4 x under the bar (mean L2 4.2e-4, 1.4 % of the pixels beyond
    1e-2) where every other config sat three orders under it, and the same frame at 64 spp from the reference build (oracle/
    gen_frames.py c4_64: one hour of CPU) showed what that was: unbiased noise of about one sample in a thousand that took another
    path than the reference's -- 4.19e-4 -> 3.95e-4, worst pixel 0.114 -> 0.034, frame means equal to 1e-6 (profiles/sessions/
    r05_c4_before_libm.txt).  Its cause was the last bit of sinf / cosf: a sampled direction one ulp off moves a hit point 1e-5
    across one of the 4000 prisms' silhouettes.  With the C library's own sinf / cosf / acosf / tanf / powf on the device
    (csrc/device_libm.h) both frames are the reference's to 1e-8 and no pixel differs by 1e-5."""
    from tuturenderer_amd import scenes

    for name, tag in (("c4", "16 spp"), ("c4_64", "64 spp")):
        f, w = _frame(tr, name, lambda: scenes.broom_room(1600, 900))
        _compare(f"c4 broom stand-in {tag}", f, w)


EXACT_SUM_FRAMES = {
    "c1": lambda sc: sc.cornell_box(800, 800), "c2": lambda sc: sc.cornell_box(800, 800), "c5": lambda sc: sc.veach_room(800, 600, small_light=False),
    "c3": lambda sc: sc.bunny_box(1024, 1024), "c4": lambda sc: sc.broom_room(1600, 900), "c4_64": lambda sc: sc.broom_room(1600, 900),
}


@pytest.mark.parametrize("name", list(EXACT_SUM_FRAMES))
def test_exact_sum_frame_is_the_reference_builds_bit_for_bit(tr, name):
    """Round 5, knob exact_sum (TUTU_EXACT_SUM=1): a path's radiance is folded from its deepest vertex back -- L_v = S_v + L_{v+1} * coe_v,
    the order in which the reference's recursion returns (PathTracing.hpp:275-277, :133) -- instead of carried forward as a sum of
    throughput * term.  With the C library's functions restated for the device (csrc/device_libm.h) nothing else differed: every
    BASELINE frame is then the reference build's frame, all 0.64-1.44 M pixels of it, bit for bit (c2: 0.33 G samples; the knob costs
    4-13 % of the frame rate, DESIGN.md 0 item 10, so the default keeps the forward sum and differs by the rounding of that sum alone: 1e-8)."""
    from tuturenderer_amd import scenes

    z = np.load(golden_path(f"frame_{name}.npz"))
    sc = EXACT_SUM_FRAMES[name](scenes)
    with tr.Context(sc) as ctx:
        ctx.set_option("exact_sum", 1)
        frame = ctx.render(int(z["spp"]), int(z["key0"]), int(z["key1"]))
        few = ctx.render(int(z["spp"]), int(z["key0"]), int(z["key1"]), max_paths=3_000_000) if name == "c1" else None
    want = z["rgb"]
    n_diff = int((frame.view(np.uint32) != want.view(np.uint32)).any(-1).sum())
    print(f"\n[{name}, exact_sum] pixels whose bits differ from the reference build's: {n_diff} of {want.shape[0] * want.shape[1]}")
    assert n_diff == 0
    if few is not None:  # (many small passes: the same frame)
        assert (few.view(np.uint32) == want.view(np.uint32)).all()


NATIVE_FRAMES = {  # fixture -> (scene maker, tile size, bound on max |z|, bounds on mean z^2)
    "native_cornell": (lambda s: s.cornell_box(128, 128), 16, 5.0, (0.5, 1.8)),
    "native_veach": (lambda s: s.veach_room(160, 120, small_light=False), 8, 5.5, (0.5, 1.8)),
    # (rough glass: a heavy-tailed estimator -- the variance of a tile mean, itself estimated from 64 replicas, is loose)
    "native_ggxT_mirror": (lambda s: __import__("oracle.gen_frames", fromlist=["_ggx_mirror"])._ggx_mirror(), 16, 7.0, (0.4, 2.5)),
}


@pytest.mark.parametrize("name", sorted(NATIVE_FRAMES))
def test_native_rng_statistical_pin(tr, name):
    """SURVEY.md 8d parity (ii): the reference with its OWN random numbers (thread_local std::mt19937 seeded from
    std::random_device, global.hpp:182-199; oracle/_ref/libtutu_ref_native.so, no engine swap) against the device with Philox
    streams -- two independent Monte-Carlo estimates of the same picture: the Cornell box at 4096 spp, the veach room and the
    rough-glass / mirror Cornell variant at 2048.  z-test of the tile means: the variance of a tile mean is measured on the
    device from 64 independent frames of spp / 64 (other Philox keys) and taken to be the same for both estimators."""
    from tuturenderer_amd import scenes

    mk, T, zmax, (chi_lo, chi_hi) = NATIVE_FRAMES[name]
    z = np.load(golden_path(f"frame_{name}.npz"))
    want = z["rgb"]
    spp = int(z["spp"])
    sc = mk(scenes)
    H, W = want.shape[:2]
    assert (sc["height"], sc["width"]) == (H, W) and H % T == 0 and W % T == 0
    assert pc.checksum(np.ascontiguousarray(sc["verts"], np.float32), np.ascontiguousarray(sc["normals"], np.float32),
                       np.ascontiguousarray(sc["mat_id"], np.int32)) == z["scene_crc"]
    tiles = lambda img: img.reshape(H // T, T, W // T, T, 3).mean(axis=(1, 3))
    with tr.Context(sc) as ctx:
        frame = ctx.render(spp, pc.KEY0, 777)
        reps = np.stack([tiles(ctx.render(spp // 64, pc.KEY0, 1000 + k)) for k in range(64)])
    var_tile = reps.var(axis=0, ddof=1) / 64.0          # variance of a tile mean at `spp` samples per pixel
    zt = (tiles(frame) - tiles(want)) / np.sqrt(2.0 * var_tile + 1e-12)
    chi = float((zt ** 2).mean())
    print(f"\n[native RNG, {name}] frame means device {frame.mean():.5f} / reference {want.mean():.5f}; tile z: max |z| {np.abs(zt).max():.2f}, "
          f"mean z^2 {chi:.2f} over {zt.size} values")
    for ch in range(3):
        # per-channel image mean within 0.5 %, plus four standard errors of the two estimates where that is more
        se = float(np.sqrt(2.0 * var_tile[..., ch].mean() / var_tile[..., ch].size))
        assert abs(frame[..., ch].mean() - want[..., ch].mean()) < 0.005 * want[..., ch].mean() + 4 * se
    assert np.abs(zt).max() < zmax
    assert chi_lo < chi < chi_hi
