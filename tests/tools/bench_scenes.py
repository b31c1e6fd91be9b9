#!/usr/bin/env python3
"""Throughput + matched-seed parity spot check on every BASELINE config scene (not the contract bench: see bench.py).
Lives under tests/ because it calls the CPU oracle as a checker (400 samples per scene).
   python tests/tools/bench_scenes.py [scene ...]   scenes: cornell veach bunny broom cornell_textured"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tuturenderer_amd as tr  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402
from tuturenderer_amd import scenes  # noqa: E402

CFG = {
    "cornell": (lambda: scenes.cornell_box(800, 800), 2, 64),
    "veach": (lambda: scenes.veach_room(800, 600), 5, 32),
    "veach_slight": (lambda: scenes.veach_room(800, 600, small_light=True), 6, 32),
    "bunny": (lambda: scenes.bunny_box(1024, 1024), 3, 16),
    "broom": (lambda: scenes.broom_room(1600, 900), 4, 4),
    "cornell_textured": (lambda: scenes.cornell_textured(800, 800), 11, 64),
    "cornell_spheres": (lambda: scenes.cornell_spheres(800, 800), 12, 64),
}


def main():
    names = sys.argv[1:] or ["cornell", "veach", "bunny", "broom"]
    P = Oracle("port")
    for name in names:
        mk, key1, spp = CFG[name]
        sc = mk()
        t0 = time.perf_counter()
        ctx = tr.Context(sc)
        t_create = time.perf_counter() - t0
        info = ctx.info()
        ctx.render(1, 0x5EED0001, key1, full_frame=False)  # warm-up / allocation
        t0 = time.perf_counter()
        img = ctx.render(spp, 0x5EED0001, key1)
        dt = time.perf_counter() - t0
        st = ctx.last_stats
        # parity spot check: per-sample radiance against the CPU restatement
        rng = np.random.default_rng(11)
        n = 400
        pix = rng.integers(0, ctx.W * ctx.H, n).astype(np.uint32)
        smp = rng.integers(0, spp, n).astype(np.uint32)
        L = ctx.trace_samples(pix, smp, 0x5EED0001, key1)
        S = P.scene(sc)
        t0 = time.perf_counter()
        Lr = S.trace_samples(pix, smp, 0x5EED0001, key1)
        t_cpu = time.perf_counter() - t0
        S.close()
        err = np.abs(L - Lr).max(1)
        scale = np.maximum(np.abs(Lr).max(1), 1e-3)
        bad = int((err > 1e-4 * scale + 1e-6).sum())
        print(json.dumps({"scene": name, "tris": info["n_tris"], "bvh_depth": info["depth"], "spp": spp,
                          "Msamples_per_s": round(ctx.W * ctx.H * spp / dt / 1e6, 1), "seconds": round(dt, 3),
                          "create_s": round(t_create, 3), "mean": float(img.mean()), "nan_pixels": int(np.isnan(img).any(-1).sum()),
                          "ms": {k: round(v, 1) for k, v in st.items() if k.startswith("ms_")},
                          "rays_per_sample": round((st["closest_rays"] + st["shadow_rays"]) / st["samples"], 2),
                          "N_T_closest": [round((st["nodes_closest"] + st["leaves_closest"]) / max(st["closest_rays"] - ctx.W * ctx.H, 1), 1),
                                          round(st["leaves_closest"] / max(st["closest_rays"] - ctx.W * ctx.H, 1), 2)],
                          "N_T_shadow": [round((st["nodes_any"] + st["leaves_any"]) / max(st["shadow_rays"], 1), 1),
                                         round(st["leaves_any"] / max(st["shadow_rays"], 1), 2)],
                          "parity_bad_of_400": bad, "cpu_port_us_per_sample": round(t_cpu / n * 1e6, 1)}), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
