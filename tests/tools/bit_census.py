"""GPU diagnostic (not a test): where the HIP path's float32 results still differ from the reference build's IN THE LAST BITS.
Prints, per material function output and per golden scene's per-sample radiances, the share of values whose bits differ and the
largest difference in ulps.  Round 5 used it after csrc/device_libm.h to find which operation is left."""
import sys

import numpy as np

sys.path.insert(0, "tests")
from conftest import golden_path  # noqa: E402
from helpers import ulp_diff  # noqa: E402
from oracle import parity_cases as pc  # noqa: E402
from oracle.gen_golden import golden_scenes  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402

import tuturenderer_amd as tr  # noqa: E402
from tuturenderer_amd import scenes  # noqa: E402

tr.load_library()
port = Oracle("port")


class Dev:
    def __init__(self, ctx):
        self.ctx = ctx

    def normalized(self, v):
        return port.normalized(v)

    def reflect(self, a, b):
        return port.reflect(a, b)

    def refract(self, a, b, c, d):
        return port.refract(a, b, c, d)

    def mat_bxdf(self, *a, **k):
        return self.ctx.eval_bxdf(*a, **k)

    def mat_pdf(self, *a, **k):
        return self.ctx.eval_pdf(*a, **k)

    def mat_sample(self, *a, **k):
        return self.ctx.eval_sample(*a, **k)


zf = np.load(golden_path("functions.npz"))
with tr.Context(scenes.cornell_box(8, 8)) as ctx:
    for name, mat in pc.material_set():
        want = {k[len(name) + 1:]: zf[k] for k in zf.files if k.startswith(name + ".")}
        got = pc.run_material(Dev(ctx), name, mat, s_wi_at=want["s_wi"])
        got = {k[len(name) + 1:]: v for k, v in got.items()}
        for k in ("bxdf", "bxdf_tir", "pdf", "s_wi", "s_pdf", "s_bxdf"):
            g, w = np.asarray(got[k], np.float32), np.asarray(want[k], np.float32)
            fin = np.isfinite(g) & np.isfinite(w)
            u = ulp_diff(g[fin], w[fin])
            nz = (g.view(np.uint32) != w.view(np.uint32)) & ~(np.isnan(g) & np.isnan(w))
            print(f"fn {name:24s} {k:9s} n {g.size:6d} differ {int(nz.sum()):6d} ({nz.mean():.5f}) max ulp {int(u.max()) if u.size else 0}")

for name in sys.argv[1:] or ["cornell", "cornell_ggxT_mirror", "cornell_ggxR_glass", "veach", "cornell_textured", "cornell_spheres"]:
    mk, key1 = golden_scenes()[name]
    sc = mk()
    z = np.load(golden_path(f"scene_{name}.npz"))
    S = port.scene(sc)
    pix, smp = pc.sample_ids(S)
    with tr.Context(sc) as ctx:
        L = ctx.trace_samples(pix, smp, pc.KEY0, key1)
    want = z["samples.L"]
    both_nan = np.isnan(L).any(1) & np.isnan(want).any(1)
    diff = (L.view(np.uint32) != want.view(np.uint32)).any(1) & ~both_nan
    Lp = port_L = None
    try:
        Lp = S.trace_samples(pix, smp, pc.KEY0, key1)
    except Exception as e:  # noqa: BLE001
        print("port trace_samples unavailable:", e)
    msg = f"samples {name:22s} n {len(L)} bits differ from the reference build: {int(diff.sum())} ({diff.mean():.4f})"
    if Lp is not None:
        dp = (Lp.view(np.uint32) != want.view(np.uint32)).any(1) & ~(np.isnan(Lp).any(1) & np.isnan(want).any(1))
        msg += f"; the C port differs in {int(dp.sum())}"
    print(msg)
    idx = np.flatnonzero(diff)[:6]
    for i in idx:
        print("   ", int(pix[i]), int(smp[i]), L[i], want[i])
    S.close()
