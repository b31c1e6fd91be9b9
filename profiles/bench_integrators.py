#!/usr/bin/env python3
"""Throughput of the OTHER integrators behind the seam (LightTracing / NaivePT / BDPT, tutu_hip_render_integrator) next to
the reference's own integrators on one host thread -- a widening row, not the headline: bench.py stays the
PathTracing benchmark.  One JSON line per integrator.

    python profiles/bench_integrators.py [--width 800 --height 800 --spp 16 --steps 3] [--no-cpu]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--scene", default="cornell_box", choices=["cornell_box", "veach_room"])
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    import numpy as np
    import tuturenderer_amd as tr
    from tuturenderer_amd import scenes

    sc = getattr(scenes, args.scene)(args.width, args.height)
    cpu = {}
    if not args.no_cpu:
        from oracle.pyoracle import Oracle, available

        kind = "reference_fast" if available("reference_fast") else ("reference" if available("reference") else "port")
        small = getattr(scenes, args.scene)(160, 120)
        S = Oracle(kind).scene(small)
        for name, it in (("light", 1), ("naivept", 2), ("bdpt", 3)):
            spp = 4
            t0 = time.perf_counter()
            S.render_integrator(it, spp, 1, 2)
            dt = time.perf_counter() - t0
            cpu[name] = {"value": 160 * 120 * spp / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": kind,
                         "sample": f"160x120, {spp} spp, one thread, {dt:.2f} s (the reference splits rows over 20 threads)"}
        S.close()
    with tr.Context(sc) as ctx:
        for name in ("light", "naivept", "bdpt"):
            ctx.render_integrator(name, args.spp, 1, 2)  # warm-up: buffers
            ms = []
            t0 = time.perf_counter()
            for _ in range(args.steps):
                img = ctx.render_integrator(name, args.spp, 1, 2)
                ms.append(ctx.last_stats["ms_total"])
            wall = (time.perf_counter() - t0) / args.steps
            n = args.width * args.height * args.spp
            print(json.dumps({"integrator": name, "metric": "Msamples/sec", "value": n / wall / 1e6, "unit": "Msamples/s", "ms_per_frame_wall": wall * 1e3,
                              "ms_per_frame_device": float(np.mean(ms)), "batches": ctx.last_stats["passes"],
                              "config": {"workload": f"{args.scene} {args.width}x{args.height}, {args.spp} spp", "frame_mean": float(img.mean())},
                              "cpu_baseline": cpu.get(name)}), flush=True)


if __name__ == "__main__":
    main()
