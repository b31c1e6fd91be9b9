#!/usr/bin/env python3
"""run-to-run determinism of a frame: render the same frame several times, compare bit for bit, print what differs"""
import os, sys, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tuturenderer_amd as tr
from tuturenderer_amd import scenes
import bench

cfgname = sys.argv[1] if len(sys.argv) > 1 else "c4"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
cfg = bench.configs()[cfgname]
scene = cfg["mk"]()
ctx = tr.Context(scene, device=0)
ref = None
for k in range(reps):
    img = ctx.render(spp, bench.KEY0, cfg["key1"])
    crc = zlib.crc32(img.tobytes()) & 0xFFFFFFFF
    if ref is None:
        ref = img
        print(f"{cfgname} spp {spp} run {k}: crc {crc:08x}", flush=True)
        continue
    d = np.argwhere((img.view(np.uint32) != ref.view(np.uint32)).any(axis=2))
    print(f"{cfgname} spp {spp} run {k}: crc {crc:08x} differing pixels {len(d)}", flush=True)
    for (y, x) in d[:8]:
        print("    pixel", int(x), int(y), "ref", ref[y, x], "now", img[y, x], "delta*spp", (img[y, x].astype(np.float64) - ref[y, x]) * spp, flush=True)
ctx.close()
